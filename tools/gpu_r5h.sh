#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h2
mkdir -p $O
timeout 600 python -m pytest tests/test_split_tower.py -m gpu -x -q -k "conv_error or aliasing or resblock" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/status.txt
tail -3 $O/tests.log
: > $O/split_bench.txt
for i in 1 2 3; do
  timeout 200 python tools/split_bench.py 2>/dev/null | grep "^split " | grep ms >> $O/split_bench.txt
  AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_RB_FULL.so timeout 200 python tools/split_bench.py 2>/dev/null | grep "^split " | grep ms | sed 's/^split/OLD  /' >> $O/split_bench.txt
done
CONV_BENCH_SHAPE=9,64 timeout 200 python tools/split_bench.py 2>/dev/null | grep "^split " | grep ms >> $O/split_bench.txt
CONV_BENCH_SHAPE=9,64 AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_RB_FULL.so timeout 200 python tools/split_bench.py 2>/dev/null | grep "^split " | grep ms | sed 's/^split/OLD64/' >> $O/split_bench.txt
cat $O/status.txt $O/split_bench.txt
