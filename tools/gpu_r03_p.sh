cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > $O/r03p_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03p_pytest.log; tail -4 $O/r03p_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/soak.py 2000 gomoku 13 > $O/r03p_soak_gomoku13_2000rounds.json 2> $O/r03p_soak_gomoku.err; tail -c 600 $O/r03p_soak_gomoku13_2000rounds.json; tail -3 $O/r03p_soak_gomoku.err
timeout 900 python bench.py --sims 400 --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --cpu-seconds 15 > $O/r03p_bench_go9_s400_c4_1gpu.json 2> $O/r03p_bench_c4.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03p_bench_driver_cmd.json 2> $O/r03p_bench_driver_cmd.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03p_bench_go9_s400_c4_1gpu.json","r03p_bench_driver_cmd.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["sims_per_move"], d["fp32_moves_per_s"], d["fresh_tree_moves_per_s"], d["speedup_vs_cpu_baseline"], d["per_rank"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
