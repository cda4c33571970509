cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/concurrency_probe2.py > $O/r03h_concurrency_probe2.txt 2>&1; tail -2 $O/r03h_concurrency_probe2.txt | cut -c1-1200
timeout 600 python tools/overlap_debug.py go 9 128 > $O/r03h_overlap_debug.txt 2>&1; tail -5 $O/r03h_overlap_debug.txt | cut -c1-400
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "range_rounds or parallel_evaluation" > $O/r03h_pytest.log 2>&1; tail -4 $O/r03h_pytest.log
