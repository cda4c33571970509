cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > $O/r03m_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03m_pytest.log; tail -5 $O/r03m_pytest.log
cat $O/precision_parity_arena_gomoku13.json $O/precision_go19_20x256_full_depth.json
timeout 900 python tools/soak.py 4000 > $O/r03m_soak_go9_4000rounds.json 2> $O/r03m_soak.err; tail -c 1200 $O/r03m_soak_go9_4000rounds.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03m_bench_driver_cmd.json 2> $O/r03m_bench_driver_cmd.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03m_bench_driver_cmd.json",):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["nn_roofline"]["avg_forward_ms"], d["fp32_moves_per_s"], d["fresh_tree_moves_per_s"], d["speedup_vs_cpu_baseline"], d["per_rank"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
