# Round 3, GPU call B: full GPU tier (fused block with the shared last tile, two half-batch streams, single-rank RCCL), probe
# variants of the fused block kernel, PMC passes on it, C2 bench and C3 bench with / without the overlap.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/r03b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03b_pytest.log
tail -8 $O/r03b_pytest.log
P=$GRAFT_REPO_ROOT/tools/probes/block64_probe
for mode in 2 1 0; do timeout 120 $P 32768 $mode 17; done > $O/r03b_block64_probe.txt 2>&1
timeout 120 $P 32768 2 9 >> $O/r03b_block64_probe.txt 2>&1
cat $O/r03b_block64_probe.txt
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --cpu-seconds 10 > $O/r03b_bench_gomoku13_c2.json 2> $O/r03b_bench_gomoku13_c2.err; python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03b_bench_gomoku13_c2.json",):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["overlap"], d["nn_roofline"]["avg_forward_ms"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python bench.py --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03b_bench_c3_overlap.json 2> $O/r03b_bench_c3_overlap.err
timeout 600 python bench.py --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline --no-overlap > $O/r03b_bench_c3_serial.json 2> $O/r03b_bench_c3_serial.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03b_bench_c3_overlap.json","r03b_bench_c3_serial.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["overlap"], d["nn_roofline"]["avg_forward_ms"], d["engine_roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
: > $O/r03b_pmc_block64.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- $P 32768 2 17 > /tmp/pb.log 2>&1
  echo "== $C" >> $O/r03b_pmc_block64.txt
  python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/r03b_pmc_block64.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%resblock64%' or kernel_name like '%conv3x3_t64%' group by kernel_name, counter_name"):
    print("  ", r[0][:60], r[1], r[2], f"{r[3]:.6g}")
PY
done
cat $O/r03b_pmc_block64.txt
