# On the GPU box: HBM traffic of the select kernel (separate --pmc passes, KiB per dispatch) for profiles/select_kernel_pmc.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d /tmp/s_$C -- python $R/bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline --no-fp32 > /tmp/s_$C.log 2>&1
  echo "== $C (bench.py --steps 30 --warmup 5 --no-graph, after the 300-round pre-roll)" >> $R/gpurun_out/select_pmc.txt
  python $R/tools/pmc_summary.py $(find /tmp/s_$C -name "*.db" | head -1) "%OpSelect%" >> $R/gpurun_out/select_pmc.txt
  python $R/tools/pmc_summary.py $(find /tmp/s_$C -name "*.db" | head -1) "%OpBackup%" >> $R/gpurun_out/select_pmc.txt
done
cat $R/gpurun_out/select_pmc.txt
