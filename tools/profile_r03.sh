# Round 3 profile run on the GPU box (what profiles/r03_* come from): same-box A/B of the 19x19 kernel (frozen round-2 copy vs the product), whole GPU tier, the driver-command and C2 benches, rocprofv3 kernel stats, PMC passes over the fused block.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
P=$GRAFT_REPO_ROOT/tools/probes
: > $O/r03l_conv19_ab.txt
for rep in 1 2; do for v in 0 2 1; do
  echo -n "round-2 kernel: " >> $O/r03l_conv19_ab.txt; timeout 60 $P/conv19_probe_FULL 4096 2 $v >> $O/r03l_conv19_ab.txt 2>&1
  echo -n "round-3 kernel: " >> $O/r03l_conv19_ab.txt; timeout 60 $P/conv19_probe_NEW 4096 2 $v >> $O/r03l_conv19_ab.txt 2>&1
done; done
cat $O/r03l_conv19_ab.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/r03l_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03l_pytest.log; tail -4 $O/r03l_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03l_bench_driver_cmd.json 2> $O/r03l_bench_driver_cmd.err
timeout 900 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --cpu-seconds 20 > $O/r03l_bench_gomoku13_c2.json 2> $O/r03l_bench_gomoku13_c2.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
for f in ("r03l_bench_driver_cmd.json","r03l_bench_gomoku13_c2.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["nn_roofline"]["avg_forward_ms"], d["fp32_moves_per_s"], d["fresh_tree_moves_per_s"], d["speedup_vs_cpu_baseline"], d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
    except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03l_bench_under_rocprof_graph.json 2> /tmp/kt1.err
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt1 -name "*.db" | head -1) > $O/r03l_kernel_stats_default_graph.txt 2>&1; head -12 $O/r03l_kernel_stats_default_graph.txt | cut -c1-150
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 40 --warmup 10 --no-fp32 --no-fresh-tree --no-cpu-baseline > $O/r03l_bench_gomoku13_under_rocprof.json 2> /tmp/kt2.err
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $O/r03l_kernel_stats_gomoku13_c2_graph.txt 2>&1; head -10 $O/r03l_kernel_stats_gomoku13_c2_graph.txt | cut -c1-150
PB=$GRAFT_REPO_ROOT/tools/probes/block64_probe
: > $O/r03l_pmc_block64.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- $PB 32768 2 17 > /tmp/pb.log 2>&1
  echo "== $C" >> $O/r03l_pmc_block64.txt
  python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/r03l_pmc_block64.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%resblock64%, 4>%' or kernel_name like '%conv3x3_t64%' group by kernel_name, counter_name"):
    print("  ", r[0][:60], r[1], r[2], f"{r[3]:.6g}")
PY
done
cat $O/r03l_pmc_block64.txt
