#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run44; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_build.py -x -q -k "valid_graphs" > $O/pytest_default.log 2>&1; tail -15 $O/pytest_default.log | cut -c1-300
FNV_FUZZ_SEED=3 FNV_FUZZ_TRIALS=80 timeout 2400 python -m pytest tests/test_gpu_device_build.py -x -q -k "valid_graphs" > $O/pytest_deep.log 2>&1; tail -15 $O/pytest_deep.log | cut -c1-300
