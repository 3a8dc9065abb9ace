#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run43; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_build.py -x -q > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log | cut -c1-300
for seed in 5 6; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=150 timeout 2400 python -m pytest tests/test_gpu_device_build.py -x -q -k "random_shapes" > $O/pytest_deep$seed.log 2>&1; grep "AssertionError: trial\|passed\|failed" $O/pytest_deep$seed.log | cut -c1-250
done
