#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run62; mkdir -p $O
for vs in 0 3072 4096 6144 8192; do
  for kind in sift glove sift_u8; do
  timeout 600 python tools/occ_probe.py $kind 52,64,100,128 visited_slots=$vs 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/vs$vs /" | tee -a $O/occ.txt
  done
done
