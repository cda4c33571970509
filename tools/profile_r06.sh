# Round 6 PMC evidence.  On the GPU box: PMC_FAMILIES="split9 hb19" bash tools/profile_r06.sh ; outputs under gpurun_out/r6p/
#   one rocprofv3 run per counter group (the guide's recipe: counters in their own runs, never with a trace domain), then
#   tools/pmc_to_json.py refreshes the family's summary (a copy of profiles/<file>, updated) next to the text.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
mkdir -p $O
declare -A JS=([split9]=split_kernel_pmc.json [split17]=split17_kernel_pmc.json [splitblock17]=splitblock17_kernel_pmc.json [hb19]=conv19_kernel_pmc.json \
               [tiled9]=conv_kernel_pmc.json [splitblock9_64]=splitblock9_64_kernel_pmc.json [split9_64]=split9_64_kernel_pmc.json [spg19]=spg19_kernel_pmc.json)
for FAM in ${PMC_FAMILIES:-split9}; do
  T=$O/r06_pmc_$FAM.txt
  : > $T
  rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/pmc_launches.py $FAM > /tmp/kt.log 2>&1
  echo "== kernel-trace" >> $T
  python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | grep -E "conv3x3|resblock|calls" | head -6 | cut -c1-220 >> $T
  for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- python $R/tools/pmc_launches.py $FAM > /tmp/pb.log 2>&1
    echo "== $C" >> $T
    python - "$(find /tmp/pb -name '*.db' | head -1)" >> $T 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%conv3x3%' or kernel_name like '%resblock%' group by kernel_name, counter_name"):
    print("  ", r[0][:70], r[1], "n=%d" % r[2], "mean=%.6g min=%.6g max=%.6g" % (r[3], r[4], r[5]))
PY
  done
  cp $R/profiles/${JS[$FAM]} $O/${JS[$FAM]} 2>/dev/null
  python $R/tools/pmc_to_json.py $FAM $T $O/${JS[$FAM]}
done
