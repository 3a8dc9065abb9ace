#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run46; mkdir -p $O
for seed in 101 102 103 104 105; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=600 FNV_FUZZ_ORACLE_EVERY=4 timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/search_$seed.log 2>&1; echo "search seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/search_$seed.log | cut -c1-250)"
done
for seed in 201 202 203; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=200 timeout 1800 python -m pytest tests/test_gpu_device_build.py -x -q -k "sequential_device_insertion_on_random" > $O/seq_$seed.log 2>&1; echo "sequential build seed $seed: $(grep 'AssertionError: trial\|passed\|failed' $O/seq_$seed.log | cut -c1-250)"
done
for seed in 301 302; do
FNV_FUZZ_SEED=$seed FNV_FUZZ_TRIALS=60 timeout 1800 python -m pytest tests/test_gpu_device_build.py -x -q -k "valid_graphs" > $O/batched_$seed.log 2>&1; echo "batched build seed $seed: $(grep 'AssertionError\|passed\|failed' $O/batched_$seed.log | cut -c1-250)"
done
