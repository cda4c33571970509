# On the GPU box: bench line, rocprofv3 kernel stats of the same command (graph off so kernels are visible one by one),
# and the conv kernel's HBM traffic (separate --pmc passes, FETCH_SIZE / WRITE_SIZE in KiB).  Text summaries -> gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py --steps 100 --warmup 60 2>&1 | tail -1 > $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 40 --warmup 60 --no-graph --no-cpu-baseline > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log > $O/bench_nograph_under_rocprof.json
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d /tmp/p_$C -- python $R/tools/conv_bench.py > /tmp/p_$C.log 2>&1
  echo "== $C" >> $O/conv_pmc.txt
  python $R/tools/pmc_summary.py $(find /tmp/p_$C -name "*.db" | head -1) "%conv3x3_tiled%" >> $O/conv_pmc.txt
  python - "$(find /tmp/p_$C -name '*.db' | head -1)" >> $O/conv_pmc.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_tiled%' group by kernel_name, counter_name"):
    print("  ", r[0][:60], r[1], r[2], f"{r[3]:.6g}")
PY
done
cat $O/bench_default.json; head -12 $O/kernel_stats.txt | cut -c1-160; cat $O/conv_pmc.txt
