#!/bin/bash
# round-5 GPU call A: new range-safety tests + affected suites, the MFMA power probe, driver-command bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_range_safety.py tests/test_split_tower.py tests/test_network.py -m gpu -x -q > gpurun_out/r5a/tests_a.log 2>&1
echo "tests_a rc=$?" >> gpurun_out/r5a/status.txt
timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "full_size_c2" > gpurun_out/r5a/tests_c2.log 2>&1
echo "tests_c2 rc=$?" >> gpurun_out/r5a/status.txt
timeout 120 ./tools/probes/mfma_power_probe gpurun_out/r5a/mfma_power_probe.json > gpurun_out/r5a/mfma_power_probe.txt 2>&1
echo "probe rc=$?" >> gpurun_out/r5a/status.txt
cp gpurun_out/r5a/mfma_power_probe.json profiles/mfma_power_probe.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
echo "bench rc=$?" >> gpurun_out/r5a/status.txt
tail -c 600 gpurun_out/r5a/tests_a.log; tail -c 400 gpurun_out/r5a/tests_c2.log; cat gpurun_out/r5a/mfma_power_probe.txt; cat gpurun_out/r5a/status.txt; tail -c 1500 gpurun_out/r5a/bench.err
