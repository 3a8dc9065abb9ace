#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run40; mkdir -p $O
for seed in 1 2 3 4 5 6; do
  mkdir -p $O/s$seed
  timeout 600 python tools/fuzz_debug.py $O/s$seed $seed 500 2>&1 | grep -v amdgpu | tee -a $O/fuzz.txt
done
