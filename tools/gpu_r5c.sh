#!/bin/bash
# round-5 GPU call C: PMC passes fused vs unfused 17x17 kernels, 19x19 map change tests + conflicts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/ -m gpu -x -q -k "19 or hb19 or go19 or c5 or jumbo" > gpurun_out/r5c/tests_19.log 2>&1
echo "tests_19 rc=$?" >> gpurun_out/r5c/status.txt
tail -3 gpurun_out/r5c/tests_19.log
PMC_FAMILIES="splitblock17 split17" bash tools/profile_r05.sh > gpurun_out/r5c/profile.log 2>&1
# 19x19: LDS conflicts only
cd /tmp
rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pb -- python $GRAFT_REPO_ROOT/tools/pmc_launches.py hb19 > /tmp/pb.log 2>&1
python - "$(find /tmp/pb -name '*.db' | head -1)" > $GRAFT_REPO_ROOT/gpurun_out/r5c/pmc_hb19.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%conv3x3%' group by kernel_name, counter_name"):
    print("  ", r[0][:70], r[1], "n=%d" % r[2], "mean=%.6g min=%.6g max=%.6g" % (r[3], r[4], r[5]))
PY
cd $GRAFT_REPO_ROOT
cat gpurun_out/r5c/status.txt; cat gpurun_out/r5p/pmc_splitblock17.txt | cut -c1-200; cat gpurun_out/r5p/pmc_split17.txt | cut -c1-200; cat gpurun_out/r5c/pmc_hb19.txt | cut -c1-200
