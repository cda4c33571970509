#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g
mkdir -p $O
timeout 300 python -m pytest tests/test_split_tower.py -m gpu -x -q -k "resblock17" > $O/tests_block.log 2>&1; echo "tests rc=$?" > $O/status.txt
tail -2 $O/tests_block.log
: > $O/rb_bench.txt
for i in 1 2 3; do RB_BENCH_LAUNCHES=60 timeout 120 python tools/rb_bench.py 2>/dev/null | tail -1 >> $O/rb_bench.txt; RB_BENCH_LAUNCHES=60 AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_RB_FULL.so timeout 120 python tools/rb_bench.py 2>/dev/null | tail -1 >> $O/rb_bench.txt; done
cat $O/status.txt $O/rb_bench.txt
