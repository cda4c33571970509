# On the GPU box: rocprofv3 evidence for the 13x13 Gomoku evaluator (BASELINE C2): kernel stats of the bench command (graph off) and
# PMC passes over tools/conv_bench.py at the tower shape (17x17 planes, 64 filters).  Text summaries -> gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt64 -- python $R/bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 40 --warmup 40 --no-graph --no-cpu-baseline > /tmp/kt64.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kt64 -name "*.db" | head -1) > $O/kernel_stats_gomoku13.txt
export CONV_BENCH_SHAPE=17,64
: > $O/conv64_pmc.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/p64; timeout 300 rocprofv3 --pmc $C -d /tmp/p64 -- python $R/tools/conv_bench.py > /tmp/p64.log 2>&1
  echo "== $C" >> $O/conv64_pmc.txt
  python - "$(find /tmp/p64 -name '*.db' | head -1)" >> $O/conv64_pmc.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_t64%' group by kernel_name, counter_name"):
    print("  ", r[0][:48], r[1], r[2], f"{r[3]:.6g}")
PY
done
head -14 $O/kernel_stats_gomoku13.txt | cut -c1-150; cat $O/conv64_pmc.txt
