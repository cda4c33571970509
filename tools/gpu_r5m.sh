#!/bin/bash
# round-5 GPU call M: final-final code -- full GPU test tier, smoke, C2 + driver-command benches
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5m
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests_gpu rc=$?" >> $O/status.txt
tail -3 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 700 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?" >> $O/status.txt
cat $O/status.txt; tail -c 300 $O/smoke.log
python - <<'P'
import json
for n in ("c2","driver"):
    try:
        d=json.loads(open(f"gpurun_out/r5m/bench_{n}.json").read().strip().splitlines()[-1])
        r=d["roofline"]; print(n, d["value"], d["ms_per_step"], r["kernel"][:22], r["avg_launch_ms"], r["frac"], d.get("evaluator_range_events"), d.get("speedup_vs_cpu_baseline"), d["config"]["evaluator"][:90])
    except Exception as e: print(n, "ERR", e)
P
tail -c 300 $O/bench_driver.err
