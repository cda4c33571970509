cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
OVERLAP_DEBUG_ROUNDS=250 OVERLAP_DEBUG_MODES=serial,seq_fwd,seq_rev timeout 900 python tools/overlap_debug.py go 9 128 > $O/r03i_overlap_debug.txt 2>&1; tail -4 $O/r03i_overlap_debug.txt | cut -c1-500
