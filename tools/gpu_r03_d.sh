# Round 3, GPU call D: whole GPU tier after the geometry / zero-fill fixes, overlap determinism, ablation of the fused block kernel
# (wall time + GRBM_GUI_ACTIVE cycles per variant), C2 / C3 benches.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/r03d_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03d_pytest.log
tail -6 $O/r03d_pytest.log
timeout 600 python tools/overlap_debug.py go 9 128 > $O/r03d_overlap_debug.txt 2>&1; tail -5 $O/r03d_overlap_debug.txt
cd /tmp && export TMPDIR=/tmp
: > $O/r03d_block64_ablation.txt
for v in FULL NO_B1 NO_M NO_B2 NO_BARRIERS NO_DMA NO_EPI NO_SKIP NO_CROSS_EPI NO_FRAG; do
  B=$GRAFT_REPO_ROOT/tools/probes/block64_abl_$v
  timeout 60 $B 32768 2 $v >> $O/r03d_block64_ablation.txt 2>&1
  rm -rf /tmp/pa; timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pa -- $B 32768 2 $v > /tmp/pa.log 2>&1
  python - "$(find /tmp/pa -name '*.db' | head -1)" >> $O/r03d_block64_ablation.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
r = {row[0]: row[1] for row in db.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%resblock64%' group by counter_name")}
cyc = r.get("GRBM_GUI_ACTIVE", 0) / 8
print(f"      cycles/launch {cyc:.4g}  parked {r.get('SQ_WAIT_ANY', 0) / max(1, r.get('SQ_WAVE_CYCLES', 1)):.3f}  mfma_busy {r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, 1024 * cyc):.3f}")
PY
done
cat $O/r03d_block64_ablation.txt
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03d_bench_gomoku13_c2.json 2> $O/r03d_bench_gomoku13_c2.err
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline --no-overlap > $O/r03d_bench_gomoku13_c2_serial.json 2> $O/r03d_bench_gomoku13_c2_serial.err
timeout 600 python bench.py --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03d_bench_c3_overlap.json 2> $O/r03d_bench_c3_overlap.err
timeout 600 python bench.py --steps 100 --warmup 20 --no-fp32 --no-fresh-tree --no-cpu-baseline --no-overlap > $O/r03d_bench_c3_serial.json 2> $O/r03d_bench_c3_serial.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03d_bench_gomoku13_c2.json","r03d_bench_gomoku13_c2_serial.json","r03d_bench_c3_overlap.json","r03d_bench_c3_serial.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["overlap"]["serial_step_ms"], d["nn_roofline"]["avg_forward_ms"], d["engine_roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
