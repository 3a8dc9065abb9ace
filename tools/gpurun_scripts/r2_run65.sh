#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run65; mkdir -p $O
for orr in 10 9 8; do
  for kind in s3 glove sift; do
  timeout 900 python tools/occ_probe.py $kind 100,160,200,256 occupancy_roomy=$orr 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/or$orr /" | tee -a $O/occ.txt
  done
done
