#!/bin/bash
# round-5 GPU call E: full GPU test tier, C2 / C5 benches (full protocol), driver-command bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests_gpu rc=$?" >> $O/status.txt
tail -3 $O/tests_gpu.log
timeout 700 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/status.txt
timeout 700 python bench.py --board 19 --sims 800 --blocks 20 --filters 256 --net-dtype bf16 --games 1024 --steps 40 --warmup 10 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" >> $O/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?" >> $O/status.txt
cat $O/status.txt
python - <<'P'
import json
for n in ("c2","c5","driver"):
    try:
        d=json.loads(open(f"gpurun_out/r5e/bench_{n}.json").read().strip().splitlines()[-1])
        r=d["roofline"]; print(n, d["value"], d["ms_per_step"], r["kernel"][:22], r["avg_launch_ms"], r["frac"], d.get("evaluator_range_events"), d.get("speedup_vs_cpu_baseline"), d.get("preroll_rounds"))
    except Exception as e: print(n, "ERR", e)
P
tail -c 400 $O/bench_c5.err; tail -c 300 $O/bench_driver.err
