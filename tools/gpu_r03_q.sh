cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/split_conv_error.jsonl
timeout 900 python -m pytest tests/test_split_tower.py -m gpu -q > $O/r03q_pytest_split.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03q_pytest_split.log; tail -5 $O/r03q_pytest_split.log | cut -c1-400
timeout 300 python tools/split_bench.py 32768 > $O/r03q_split_bench.txt 2>&1; grep "^split\|^library" $O/r03q_split_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $GRAFT_REPO_ROOT/tools/split_pmc.py 32768 > /tmp/kts.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kts -name "*.db" | head -1) > $O/r03q_kernel_stats_split.txt 2>&1; head -8 $O/r03q_kernel_stats_split.txt | cut -c1-170
