# PMC passes over tools/conv_bench.py for the conv kernels (run on the GPU box: bash tools/conv_pmc.sh [pattern] [boards])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-%conv3x3_tiled%}
NB=${2:-32768}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/cpmc$i -- python $R/tools/conv_bench.py $NB > $R/gpurun_out/cpmc$i.log 2>&1
  DB=$(find $R/gpurun_out/cpmc$i -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $DB "$PAT"
  rm -rf $R/gpurun_out/cpmc$i
done
