# Round 3, fp32-class evaluator: whole GPU tier, smoke, the driver-command bench (fp32-class + library fp32 companions), rocprofv3 kernel stats and
# PMC passes of the split-precision tower kernels (what profiles/r03_split_* and r03_bench_driver_cmd_d.json come from).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/split_conv_error.jsonl
timeout 1800 python -m pytest tests -m gpu -q > $O/r03s_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03s_pytest.log; tail -6 $O/r03s_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03s_bench_driver_cmd.json 2> $O/r03s_bench_driver_cmd.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"
f="r03s_bench_driver_cmd.json"
try:
    d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["fresh_tree_moves_per_s"], d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"]); print(json.dumps(d["fp32_companion"])); print(json.dumps(d["fp32_library_companion"]))
except Exception as e: print(f, "ERR", e, open(O+f.replace(".json",".err")).read()[-1500:])
PY
timeout 300 python tools/split_bench.py 32768 > $O/r03s_split_bench.txt 2>&1; tail -8 $O/r03s_split_bench.txt | cut -c1-400
timeout 300 python tools/split_forward_bench.py 32768 > $O/r03s_split_forward_bench.txt 2>&1; tail -1 $O/r03s_split_forward_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $GRAFT_REPO_ROOT/tools/split_pmc.py 32768 > /tmp/kts.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kts -name "*.db" | head -1) > $O/r03s_kernel_stats_split.txt 2>&1; head -8 $O/r03s_kernel_stats_split.txt | cut -c1-170
bash $GRAFT_REPO_ROOT/tools/profile_split.sh > /dev/null 2>&1; cat $O/r03_pmc_split.txt | grep -v split_layout | cut -c1-120
