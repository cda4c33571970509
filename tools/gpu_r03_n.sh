cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/concurrency_probe2.py > $O/r03n_probe2_product.txt 2>&1; tail -1 $O/r03n_probe2_product.txt | cut -c1-600
AZ_PROBE_LIB=$GRAFT_REPO_ROOT/tools/probes/libazsp_headpad.so timeout 300 python tools/concurrency_probe2.py > $O/r03n_probe2_headpad.txt 2>&1; tail -1 $O/r03n_probe2_headpad.txt | cut -c1-600
