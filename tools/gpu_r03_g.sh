cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/concurrency_probe2.py > $O/r03g_concurrency_probe2.txt 2>&1; tail -3 $O/r03g_concurrency_probe2.txt | cut -c1-1500
