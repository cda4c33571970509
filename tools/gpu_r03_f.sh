cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/overlap_debug.py go 9 128 > $O/r03f_overlap_debug.txt 2>&1; tail -6 $O/r03f_overlap_debug.txt
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_engine_gpu.py::test_gpu_two_half_batch_streams_equal_serial_rounds > $O/r03f_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03f_pytest.log
tail -6 $O/r03f_pytest.log
cat $O/precision_parity_arena_gomoku13.json $O/precision_go19_20x256_full_depth.json
