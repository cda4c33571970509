#!/bin/bash
# round-5 GPU call I: final code -- full GPU test tier, C2 + driver-command benches, rocprofv3 kernel-trace summaries of both
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests_gpu rc=$?" >> $O/status.txt
tail -3 $O/tests_gpu.log
timeout 700 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?" >> $O/status.txt
cd /tmp
rm -rf /tmp/kts; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-fresh-tree > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/rocprof_summary.py $(find /tmp/kts -name "*.db" | head -1) > $O/kernel_stats_default_graph.txt 2>&1
rm -rf /tmp/kts; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $R/bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 60 --warmup 10 --no-cpu-baseline --no-companions --no-fresh-tree > $O/bench_c2_under_rocprof.json 2> $O/bench_c2_under_rocprof.err
python $R/tools/rocprof_summary.py $(find /tmp/kts -name "*.db" | head -1) > $O/kernel_stats_gomoku13_c2.txt 2>&1
cd $R
cat $O/status.txt
python - <<'P'
import json
for n in ("c2","driver","under_rocprof","c2_under_rocprof"):
    try:
        d=json.loads(open(f"gpurun_out/r5i/bench_{n}.json").read().strip().splitlines()[-1])
        r=d["roofline"]; print(n, d["value"], d["ms_per_step"], r["kernel"][:22], r["avg_launch_ms"], r["frac"], d.get("evaluator_range_events"), d.get("speedup_vs_cpu_baseline"))
    except Exception as e: print(n, "ERR", e)
P
head -8 $O/kernel_stats_default_graph.txt | cut -c1-160; head -8 $O/kernel_stats_gomoku13_c2.txt | cut -c1-160
