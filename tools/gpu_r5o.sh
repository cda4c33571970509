#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5o
mkdir -p $O
timeout 400 python bench.py --sims 400 --steps 60 --warmup 10 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; echo "c4 rc=$?" > $O/status.txt
timeout 400 python bench.py --blocks 12 --filters 64 --steps 100 --warmup 20 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_12b64.json 2> $O/bench_12b64.err; echo "12b64 rc=$?" >> $O/status.txt
cat $O/status.txt
python - <<'P'
import json
for n in ("c4_1gpu","12b64"):
    try:
        d=json.loads(open(f"gpurun_out/r5o/bench_{n}.json").read().strip().splitlines()[-1])
        r=d["roofline"]; print(n, d["value"], d["ms_per_step"], r["kernel"][:22], r["avg_launch_ms"], r["frac"], d.get("evaluator_range_events"))
    except Exception as e: print(n, "ERR", e)
P
