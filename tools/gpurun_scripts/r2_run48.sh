#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run48; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_python_api.py -x -q -k "random_operation" > $O/pytest_default.log 2>&1; tail -3 $O/pytest_default.log | cut -c1-400
for seed in 51; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=300 timeout 2400 python -m pytest tests/test_gpu_python_api.py -x -q -k "random_operation" > $O/pytest_deep$seed.log 2>&1; grep "AssertionError: trial\|passed\|failed\|Error" $O/pytest_deep$seed.log | cut -c1-400 | head -5
done
