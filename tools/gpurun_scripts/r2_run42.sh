#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run42; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/pytest_default.log 2>&1; tail -3 $O/pytest_default.log
for seed in 31 32; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=400 FNV_FUZZ_ORACLE_EVERY=3 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/pytest_seed$seed.log 2>&1; tail -3 $O/pytest_seed$seed.log | cut -c1-300
done
