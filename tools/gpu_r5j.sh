#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j
mkdir -p $O
timeout 300 python tools/vmcnt_check.py > $O/vmcnt0_check.txt 2>&1; echo "vmcnt_check rc=$?" > $O/status.txt
tail -3 $O/vmcnt0_check.txt
: > $O/rb_dma.txt
for i in 1 2; do for V in FULL DMA_Q1 DMA_Q3 TWO_PER_STEP; do
  RB_BENCH_LAUNCHES=60 AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_RB_$V.so timeout 120 python tools/rb_bench.py 2>/dev/null | tail -1 >> $O/rb_dma.txt
done; done
cat $O/status.txt; cut -c1-130 $O/rb_dma.txt
