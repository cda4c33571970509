# On the GPU box (round 2): the driver's bench line, rocprofv3 kernel stats of the SAME command (forward replayed from the hipGraph)
# and with the graph off, and the conv kernel's HBM traffic (separate --pmc passes).  Text summaries -> gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_driver_cmd.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kg -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 > /tmp/kg.log 2>&1
tail -1 /tmp/kg.log > $O/r02_bench_under_rocprof_graph.json
python $R/tools/rocprof_summary.py $(find /tmp/kg -name "*.db" | head -1) > $O/r02_kernel_stats_default_graph.txt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-fp32 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log > $O/r02_bench_under_rocprof_nograph.json
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/r02_kernel_stats_nograph.txt
rm -f $O/r02_conv_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d /tmp/p_$C -- python $R/tools/conv_bench.py > /tmp/p_$C.log 2>&1
  echo "== $C" >> $O/r02_conv_pmc.txt
  python - "$(find /tmp/p_$C -name '*.db' | head -1)" >> $O/r02_conv_pmc.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_tiled%' group by kernel_name, counter_name"):
    print("  ", r[0][:60], r[1], r[2], f"{r[3]:.6g}")
PY
done
cut -c1-300 $O/r02_bench_driver_cmd.json; head -14 $O/r02_kernel_stats_default_graph.txt | cut -c1-170; cat $O/r02_conv_pmc.txt
