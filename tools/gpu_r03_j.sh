cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/split_debug.py > $O/r03j_split_debug.txt 2>&1; tail -8 $O/r03j_split_debug.txt | cut -c1-700
