# Round 4 evidence: (1) rocprofv3 --kernel-trace --stats of the bench command in its steady state (fp32-class evaluator = the headline),
# (2) FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (separate runs, the guide's recipe) for the split kernels and -- refreshed -- the bf16
# 9x9 kernel and the 19x19 kernel.  Run on the GPU box: bash tools/profile_r04.sh ; outputs under gpurun_out/r4p/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p
mkdir -p $O
rm -rf /tmp/kts
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-fresh-tree > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/rocprof_summary.py $(find /tmp/kts -name "*.db" | head -1) > $O/kernel_stats_default_graph.txt 2>&1
head -12 $O/kernel_stats_default_graph.txt | cut -c1-200
for FAM in ${PMC_FAMILIES:-split9 split17 tiled9 hb19}; do
  : > $O/pmc_$FAM.txt
  for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- python $R/tools/pmc_launches.py $FAM > /tmp/pb.log 2>&1
    echo "== $C" >> $O/pmc_$FAM.txt
    python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/pmc_$FAM.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
# one row per kernel (the plain and the residual variant are different template instantiations) and counter: dispatches, mean per dispatch
for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%conv3x3%' group by kernel_name, counter_name"):
    print("  ", r[0][:70], r[1], "n=%d" % r[2], "mean=%.6g min=%.6g max=%.6g" % (r[3], r[4], r[5]))
PY
  done
  cat $O/pmc_$FAM.txt | cut -c1-260
done
