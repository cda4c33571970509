cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/k5 -- python $R/bench.py --board 19 --sims 800 --blocks 20 --filters 256 --games 512 --steps 6 --warmup 2 --preroll-rounds 20 --preroll-moves 0 --no-graph --no-cpu-baseline --no-fp32 > /tmp/k5.log 2>&1
tail -1 /tmp/k5.log | cut -c1-200
python $R/tools/rocprof_summary.py $(find /tmp/k5 -name "*.db" | head -1) > $R/gpurun_out/r02_kernel_stats_go19_c5_nograph.txt
head -14 $R/gpurun_out/r02_kernel_stats_go19_c5_nograph.txt | cut -c1-150
