cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/split_conv_error.jsonl
timeout 900 python -m pytest tests/test_split_tower.py -m gpu -q -s > $O/r03q_pytest_split.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03q_pytest_split.log; tail -25 $O/r03q_pytest_split.log | cut -c1-400
timeout 300 python tools/split_bench.py 32768 > $O/r03q_split_bench.txt 2>&1; tail -9 $O/r03q_split_bench.txt | cut -c1-300
timeout 300 python tools/split_forward_bench.py 32768 > $O/r03q_split_forward_bench.txt 2>&1; tail -5 $O/r03q_split_forward_bench.txt | cut -c1-300
