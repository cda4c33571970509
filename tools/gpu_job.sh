#!/bin/bash
# One parameterised GPU-box job script (replaces the per-call tools/gpu_r5[a-o].sh of round 5).
#   gpurun --timeout 1500 -- 'bash tools/gpu_job.sh <job> [<job> ...]'      outputs under gpurun_out/r6/
# jobs: conv19_ab | tests | tests:<pytest -k expr> | bench | bench_c2 | bench_c5 | bench_12b64 | bench_c4 | probe_power | pmc:<family> | rocprof_bench | rocprof_c2 | rocprof_c5 | rocprof_dropin | resblock_ab:<S> | prev_ab | bench_ab | chunk_probe | dropin_ab | spg_ab | soak | smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6
mkdir -p $O
summ() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], r["kernel"][:24], "launch_ms", r["avg_launch_ms"], "frac", r["frac"],
          "range_events", d.get("evaluator_range_events"), "c1_dropin", d.get("c1_dropin"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
P
}
for J in "$@"; do
  case "$J" in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; tail -5 $O/gpu_tests.txt ;;
    tests:*) timeout 1800 python -m pytest tests -m gpu -x -q -k "${J#tests:}" > $O/gpu_tests_k.txt 2>&1; echo "tests -k rc=$?" | tee -a $O/status.txt; tail -15 $O/gpu_tests_k.txt ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt; tail -2 $O/smoke.txt ;;
    bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?" | tee -a $O/status.txt; summ $O/bench_driver_cmd.json ;;
    bench_c2) timeout 600 python bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_gomoku13_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" | tee -a $O/status.txt; summ $O/bench_gomoku13_c2.json ;;
    bench_c5) timeout 900 python bench.py --board 19 --games 1024 --sims 800 --blocks 20 --filters 256 --net-dtype bf16 --steps 40 --warmup 5 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_go19_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" | tee -a $O/status.txt; summ $O/bench_go19_c5.json ;;
    bench_12b64) timeout 600 python bench.py --blocks 12 --filters 64 --steps 100 --warmup 20 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_go9_12b64.json 2> $O/bench_12b64.err; echo "12b64 rc=$?" | tee -a $O/status.txt; summ $O/bench_go9_12b64.json ;;
    bench_c4) timeout 600 python bench.py --sims 400 --steps 60 --warmup 10 --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_go9_s400_c4_1gpu.json 2> $O/bench_c4.err; echo "c4 rc=$?" | tee -a $O/status.txt; summ $O/bench_go9_s400_c4_1gpu.json ;;
    probe_power) (cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o mfma_power_probe mfma_power_probe.hip) && timeout 300 ./tools/probes/mfma_power_probe $O/mfma_power_probe.json > $O/mfma_power_probe.txt 2>&1; echo "probe rc=$?" | tee -a $O/status.txt; cat $O/mfma_power_probe.txt ;;
    pmc:*) PMC_FAMILIES="${J#pmc:}" bash tools/profile_r06.sh > $O/pmc_${J#pmc:}.log 2>&1; echo "pmc ${J#pmc:} rc=$?" | tee -a $O/status.txt; tail -3 $O/pmc_${J#pmc:}.log ;;
    rocprof_bench) (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rb && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rb -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-companions --no-fresh-tree --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> /tmp/rb.err; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/rb -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/kernel_stats_default.txt 2>&1); echo "rocprof rc=$?" | tee -a $O/status.txt; head -8 $O/kernel_stats_default.txt | cut -c1-200 ;;
    resblock_ab:*) timeout 300 python tools/resblock_ab.py ${J#resblock_ab:} 2>&1 | grep -v amdgpu.ids > $O/resblock_ab_${J#resblock_ab:}.txt; echo "resblock_ab rc=$?" | tee -a $O/status.txt; cat $O/resblock_ab_${J#resblock_ab:}.txt ;;
    soak) timeout 900 python tools/soak.py 1500 go 9 4096 fp32 > $O/soak_go9_fp32class_1500rounds.json 2> $O/soak_go9.err; echo "soak go9 rc=$?" | tee -a $O/status.txt
          timeout 900 python tools/soak.py 1500 go 9 4096 fp32 12 64 > $O/soak_go9_12b64_fp32class_1500rounds.json 2> $O/soak_12b64.err; echo "soak 12b64 rc=$?" | tee -a $O/status.txt
          timeout 900 python tools/soak.py 1200 gomoku 13 4096 fp32 > $O/soak_gomoku13_fp32class_1200rounds.json 2> $O/soak_gomoku.err; echo "soak gomoku rc=$?" | tee -a $O/status.txt
          timeout 900 python tools/soak.py 600 go 19 1024 bf16 20 256 800 > $O/soak_go19_c5_bf16_600rounds.json 2> $O/soak_go19.err; echo "soak go19 rc=$?" | tee -a $O/status.txt
          tail -n 1 $O/soak_*.json | cut -c1-600 ;;
    rocprof_c2) (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rb2 && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rb2 -- python $GRAFT_REPO_ROOT/bench.py --game gomoku --board 13 --blocks 6 --filters 64 --steps 100 --warmup 20 --no-companions --no-fresh-tree --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_c2_under_rocprof.json 2> /tmp/rb2.err; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/rb2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/kernel_stats_gomoku13_c2.txt 2>&1); echo "rocprof c2 rc=$?" | tee -a $O/status.txt; head -6 $O/kernel_stats_gomoku13_c2.txt | cut -c1-160 ;;
    rocprof_c5) (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rb5 && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rb5 -- python $GRAFT_REPO_ROOT/bench.py --board 19 --games 1024 --sims 800 --blocks 20 --filters 256 --net-dtype bf16 --steps 40 --warmup 5 --no-companions --no-fresh-tree --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_c5_under_rocprof.json 2> /tmp/rb5.err; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/rb5 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/kernel_stats_go19_c5.txt 2>&1); echo "rocprof c5 rc=$?" | tee -a $O/status.txt; head -6 $O/kernel_stats_go19_c5.txt | cut -c1-160 ;;
    dropin_ab) timeout 300 python tools/dropin_resident_ab.py 2>&1 | grep -v amdgpu.ids > $O/dropin_resident_ab.txt; echo "dropin_ab rc=$?" | tee -a $O/status.txt; cat $O/dropin_resident_ab.txt ;;
    rocprof_dropin) (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rbd && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rbd -- python $GRAFT_REPO_ROOT/tools/dropin_resident_ab.py 3 > $GRAFT_REPO_ROOT/$O/dropin_ab_under_rocprof.txt 2> /tmp/rbd.err; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/rbd -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/kernel_stats_dropin_c1.txt 2>&1); echo "rocprof dropin rc=$?" | tee -a $O/status.txt; head -24 $O/kernel_stats_dropin_c1.txt | cut -c1-170 ;;
    spg_ab) timeout 600 python tools/spg_ab.py 2>&1 | grep -v amdgpu.ids > $O/spg_ab.txt; echo "spg_ab rc=$?" | tee -a $O/status.txt; head -60 $O/spg_ab.txt ;;
    prev_ab) timeout 900 python tools/split_prev_ab.py 2>&1 | grep -v amdgpu.ids > $O/split_prev_ab.txt; echo "prev_ab rc=$?" | tee -a $O/status.txt; grep -v "  round" $O/split_prev_ab.txt ;;
    chunk_probe) timeout 900 python tools/chunk_major_probe.py 2>&1 | grep -v amdgpu.ids > $O/chunk_major_probe.txt; echo "chunk_probe rc=$?" | tee -a $O/status.txt; grep -v "  round" $O/chunk_major_probe.txt ;;
    bench_ab) # the driver's workload (no companions) with the library of the tree and with tools/probes/libazsp_prev.so, alternating, on this box
          cp alpha_zero_amd/libazsp.so /tmp/libazsp_tree.so
          for R in 1 2; do for V in tree prev; do
            if [ $V = prev ]; then cp tools/probes/libazsp_prev.so alpha_zero_amd/libazsp.so; else cp /tmp/libazsp_tree.so alpha_zero_amd/libazsp.so; fi
            timeout 600 python bench.py --gpus 1 --steps 40 --warmup 5 ${BENCH_AB_ARGS:-} --no-companions --no-fresh-tree --no-cpu-baseline > $O/bench_ab_${V}_$R.json 2> $O/bench_ab.err; echo "bench_ab $V $R rc=$?" | tee -a $O/status.txt; summ $O/bench_ab_${V}_$R.json
          done; done
          cp /tmp/libazsp_tree.so alpha_zero_amd/libazsp.so ;;
    conv19_ab) timeout 600 python tools/conv19_ab.py > $O/conv19_ab.txt 2>&1; echo "conv19_ab rc=$?" | tee -a $O/status.txt; cat $O/conv19_ab.txt ;;
    *) echo "unknown job $J" ;;
  esac
done
