# Round 5 evidence.  Run on the GPU box: bash tools/profile_r05.sh ; outputs under gpurun_out/r5p/
#   PMC passes (separate runs, the guide's recipe) per kernel family in $PMC_FAMILIES: HBM bytes, matrix-pipe busy, wave-cycle breakdown, LDS conflicts,
#   effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall) next to mfma busy.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5p
mkdir -p $O
for FAM in ${PMC_FAMILIES:-splitblock17 split17}; do
  : > $O/pmc_$FAM.txt
  rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/pmc_launches.py $FAM > /tmp/kt.log 2>&1
  echo "== kernel-trace" >> $O/pmc_$FAM.txt
  python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | grep -E "conv3x3|resblock|name" | head -6 | cut -c1-220 >> $O/pmc_$FAM.txt
  for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- python $R/tools/pmc_launches.py $FAM > /tmp/pb.log 2>&1
    echo "== $C" >> $O/pmc_$FAM.txt
    python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/pmc_$FAM.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%conv3x3%' or kernel_name like '%resblock%' group by kernel_name, counter_name"):
    print("  ", r[0][:70], r[1], "n=%d" % r[2], "mean=%.6g min=%.6g max=%.6g" % (r[3], r[4], r[5]))
PY
  done
  cat $O/pmc_$FAM.txt | cut -c1-260
done
