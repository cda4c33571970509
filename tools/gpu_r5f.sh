#!/bin/bash
# round-5 GPU call F: same-box ablation of the fused split block (tools/probes/make_rb_abl.py builds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f
mkdir -p $O
: > $O/rb_ablation.txt
for V in FULL NO_DMA NO_VMWAIT NO_BAR NO_STORE NO_MWRITE NO_FRAG NO_EXPOSED FULL; do
  AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_RB_$V.so timeout 120 python tools/rb_bench.py 2>/dev/null | tail -1 >> $O/rb_ablation.txt
done
cat $O/rb_ablation.txt
