cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03t_bench_driver_cmd.json 2> $O/r03t_bench_driver_cmd.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
f="r03t_bench_driver_cmd.json"
try:
    d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["fresh_tree_moves_per_s"], d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"]); print(json.dumps(d["fp32_companion"])); print(json.dumps(d["fp32_library_companion"]))
except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
