cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/split_conv_error.jsonl
timeout 900 python -m pytest tests/test_split_tower.py -m gpu -q -s > $O/r03q_pytest_split.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03q_pytest_split.log; tail -30 $O/r03q_pytest_split.log | cut -c1-300
AZ_SP_V1=1 timeout 300 python tools/split_bench.py 32768 > $O/r03q_split_bench_v1.txt 2>&1; grep "^split " $O/r03q_split_bench_v1.txt
timeout 300 python tools/split_bench.py 32768 > $O/r03q_split_bench.txt 2>&1; tail -9 $O/r03q_split_bench.txt | cut -c1-300
bash tools/profile_split.sh
