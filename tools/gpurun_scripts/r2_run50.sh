#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run50; mkdir -p $O
for t in 0 25 50 75 100 150; do
  timeout 600 python tools/occ_probe.py sift_u8 52,100 sorted_tail_exact_pct=$t 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/t$t /" | tee -a $O/u8_tail.txt
done
for t in 0 50 75 100; do
  timeout 600 python tools/occ_probe.py sift 52,100 sorted_tail_exact_pct=$t 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/t$t /" | tee -a $O/f32_tail.txt
done
