#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run70; mkdir -p $O
for seed in 5001 5002 5003 5004; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=600 FNV_FUZZ_ORACLE_EVERY=5 timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/search_$seed.log 2>&1; echo "search seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/search_$seed.log | cut -c1-250)"
done
