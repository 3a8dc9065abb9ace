#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run58; mkdir -p $O
for ot in 13 10 8; do
  for kind in glove sift; do
  timeout 600 python tools/occ_probe.py $kind 128,160,200,256,320 occupancy_target=$ot 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/ot$ot /" | tee -a $O/occ_ot.txt
  done
done
