cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/overlap_debug.py go 9 128 > $O/r03c_overlap_debug.txt 2>&1; tail -8 $O/r03c_overlap_debug.txt
timeout 900 python -m pytest tests/test_network.py tests/test_ckpt.py tests/test_nccl_single_rank.py -m gpu -q -x > $O/r03c_pytest.log 2>&1; tail -15 $O/r03c_pytest.log
