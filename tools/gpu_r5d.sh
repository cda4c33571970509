#!/bin/bash
# round-5 GPU call D: ring-depth A/B of the fused block (wave-cycle breakdown), the same breakdown for the headline kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_split_tower.py -m gpu -x -q -k "resblock17" > $O/tests_block_r6.log 2>&1; echo "tests r6 rc=$?" >> $O/status.txt
AZSP_RB_RING=3 timeout 300 python -m pytest tests/test_split_tower.py -m gpu -x -q -k "resblock17" > $O/tests_block_r3.log 2>&1; echo "tests r3 rc=$?" >> $O/status.txt
cd /tmp
run_pmc() {  # $1 = label, $2 = family, env passes through
  : > $O/pmc_$1.txt
  rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/pmc_launches.py $2 > /tmp/kt.log 2>&1
  echo "== kernel-trace" >> $O/pmc_$1.txt
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | grep -E "conv3x3|resblock" | head -4 | cut -c1-200 >> $O/pmc_$1.txt
  for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $C -d /tmp/pb -- python $GRAFT_REPO_ROOT/tools/pmc_launches.py $2 > /tmp/pb.log 2>&1
    python - "$(find /tmp/pb -name '*.db' | head -1)" >> $O/pmc_$1.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3%' or kernel_name like '%resblock%' group by kernel_name, counter_name"):
    print("  ", r[0][:44], r[1], "n=%d" % r[2], "mean=%.6g" % r[3])
PY
  done
}
run_pmc block_r6 splitblock17
AZSP_RB_RING=3 run_pmc block_r3 splitblock17
run_pmc split9 split9
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -2 $O/tests_block_r6.log; tail -2 $O/tests_block_r3.log
for f in block_r6 block_r3 split9; do echo "#### $f"; cat $O/pmc_$f.txt | cut -c1-150; done
