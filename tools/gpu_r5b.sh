#!/bin/bash
# round-5 GPU call B: fused split block parity + remaining new tests, C2 bench fused vs unfused
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_split_tower.py -m gpu -x -q -k "resblock17" > gpurun_out/r5b/tests_block.log 2>&1
echo "tests_block rc=$?" >> gpurun_out/r5b/status.txt
timeout 900 python -m pytest tests/test_range_safety.py tests/test_split_tower.py tests/test_network.py -m gpu -q > gpurun_out/r5b/tests_a.log 2>&1
echo "tests_a rc=$?" >> gpurun_out/r5b/status.txt
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 60 --warmup 10 --no-companions --no-fresh-tree --no-cpu-baseline > gpurun_out/r5b/bench_c2_fused.json 2> gpurun_out/r5b/bench_c2_fused.err
echo "bench fused rc=$?" >> gpurun_out/r5b/status.txt
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 60 --warmup 10 --no-companions --no-fresh-tree --no-cpu-baseline --no-fused-block > gpurun_out/r5b/bench_c2_unfused.json 2> gpurun_out/r5b/bench_c2_unfused.err
echo "bench unfused rc=$?" >> gpurun_out/r5b/status.txt
tail -c 1500 gpurun_out/r5b/tests_block.log; tail -c 800 gpurun_out/r5b/tests_a.log; cat gpurun_out/r5b/status.txt
python - <<'P'
import json
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r5b/bench_c2_{n}.json").read().strip().splitlines()[-1])
        r=d["roofline"]; print(n, d["value"], d["ms_per_step"], r["kernel"][:20], r["avg_launch_ms"], r["frac"], d["evaluator_range_events"])
    except Exception as e: print(n, "ERR", e)
P
tail -c 600 gpurun_out/r5b/bench_c2_fused.err
