cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/concurrency_probe.py go > $O/r03e_concurrency_probe_go.txt 2>&1; tail -3 $O/r03e_concurrency_probe_go.txt
timeout 300 python tools/concurrency_probe.py gomoku > $O/r03e_concurrency_probe_gomoku.txt 2>&1; tail -3 $O/r03e_concurrency_probe_gomoku.txt
timeout 900 python -m pytest tests/test_network.py tests/test_ckpt.py tests/test_realnet_search.py -m gpu -q -x > $O/r03e_pytest.log 2>&1; tail -5 $O/r03e_pytest.log
cat $O/realnet_search_parity.json
P=$GRAFT_REPO_ROOT/tools/probes/block64_probe
timeout 120 $P 32768 2 17 > $O/r03e_block64_probe.txt 2>&1; timeout 120 $P 32768 2 9 >> $O/r03e_block64_probe.txt 2>&1; cat $O/r03e_block64_probe.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pb -- $P 32768 2 17 > /tmp/pb.log 2>&1
python - "$(find /tmp/pb -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%resblock64%' or kernel_name like '%t64%' group by kernel_name, counter_name"):
    print("  ", r[0][:50], r[1], f"{r[2]:.6g}")
PY
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline --no-overlap > $O/r03e_bench_gomoku13_c2.json 2> $O/r03e_bench_gomoku13_c2.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03e_bench_gomoku13_c2.json",):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["nn_roofline"]["avg_forward_ms"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
