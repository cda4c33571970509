cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_network.py tests/test_precision_parity.py tests/test_engine_gpu.py -m gpu -q -k "go19 or hb19 or 256 or range_rounds" > $O/r03k_pytest.log 2>&1; tail -5 $O/r03k_pytest.log
CONV_BENCH_SHAPE=19,256 timeout 300 python tools/conv_bench.py 4096 > $O/r03k_conv_bench_19_256.log 2>&1; head -8 $O/r03k_conv_bench_19_256.log
timeout 900 python bench.py --board 19 --sims 800 --blocks 20 --filters 256 --games 1024 --steps 30 --warmup 5 --preroll-rounds 120 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03k_bench_go19_c5.json 2> $O/r03k_bench_go19_c5.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03k_bench_go19_c5.json",):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["nn_roofline"]["avg_forward_ms"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
