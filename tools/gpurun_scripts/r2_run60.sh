#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run60; mkdir -p $O
for vs in 0 1536 3072 4096; do
  timeout 600 python tools/occ_probe.py sift 32,52,64 visited_slots=$vs sorted_tail_exact_pct=100 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/vs$vs /" | tee -a $O/occ.txt
done
