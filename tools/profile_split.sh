# PMC passes over the split-precision tower convolution (k_conv3x3_sp) on the GPU box -> gpurun_out/r03_pmc_split.txt
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
: > $O/r03_pmc_split.txt
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- python $GRAFT_REPO_ROOT/tools/split_pmc.py 32768 > /tmp/pb.log 2>&1
  echo "== $C" >> $O/r03_pmc_split.txt
  python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/r03_pmc_split.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_sp%' or kernel_name like '%split_layout%' group by kernel_name, counter_name"):
    print("  ", r[0][:60], r[1], r[2], f"{r[3]:.6g}")
PY
done
cat $O/r03_pmc_split.txt
