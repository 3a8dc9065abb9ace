#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run54; mkdir -p $O
for seed in 1001 1002 1003 1004 1005 1006 1007 1008; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=600 FNV_FUZZ_ORACLE_EVERY=5 timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/search_$seed.log 2>&1; echo "search seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/search_$seed.log | cut -c1-250)"
done
for seed in 2001 2002 2003; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=250 timeout 1800 python -m pytest tests/test_gpu_device_build.py -x -q -k "sequential_device_insertion_on_random" > $O/seq_$seed.log 2>&1; echo "sequential build seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/seq_$seed.log | cut -c1-250)"
done
for seed in 3001 3002 3003; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=400 timeout 1800 python -m pytest tests/test_gpu_python_api.py -x -q -k "random_operation" > $O/api_$seed.log 2>&1; echo "api sequences seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/api_$seed.log | cut -c1-300)"
done
