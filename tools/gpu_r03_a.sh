# Round 3, GPU call A: parity of the fused ResNetBlock kernel, C2 bench + rocprof kernel stats, engine/forward overlap probe.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03a_pytest.log
tail -5 $O/r03a_pytest.log
CONV_BENCH_SHAPE=17,64 timeout 300 python tools/conv_bench.py > $O/r03a_conv_bench_17_64.log 2>&1; cat $O/r03a_conv_bench_17_64.log
CONV_BENCH_SHAPE=9,64 timeout 300 python tools/conv_bench.py > $O/r03a_conv_bench_9_64.log 2>&1; head -4 $O/r03a_conv_bench_9_64.log
timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-fp32 --cpu-seconds 15 > $O/r03a_bench_gomoku13_c2.json 2> $O/r03a_bench_gomoku13_c2.err; tail -c 3000 $O/r03a_bench_gomoku13_c2.json
timeout 300 python tools/overlap_probe.py > $O/r03a_overlap_probe.json 2> $O/r03a_overlap_probe.err; cat $O/r03a_overlap_probe.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt64 -- python $GRAFT_REPO_ROOT/bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 40 --warmup 10 --preroll-rounds 60 --no-fp32 --no-cpu-baseline > $O/r03a_bench_gomoku13_under_rocprof.json 2> /tmp/kt64.err
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt64 -name "*.db" | head -1) > $O/r03a_kernel_stats_gomoku13_c2_graph.txt 2>&1
head -16 $O/r03a_kernel_stats_gomoku13_c2_graph.txt | cut -c1-160
