#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5l
mkdir -p $O
: > $O/sp9_ablation.txt
for i in 1 2; do for V in FULL HALF_FRAG NO_FRAG NO_DMA NO_STORE; do
  AZ_BENCH_LIB=$PWD/tools/probes/libazsp_abl_$V.so timeout 200 python tools/split_bench.py 2>/dev/null | grep "^split " | grep ms | sed "s/^split/$V/" >> $O/sp9_ablation.txt
done; done
cat $O/sp9_ablation.txt
