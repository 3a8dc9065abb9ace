#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run25; mkdir -p $O
for ot in 13 11 9 7; do
  timeout 600 python tools/occ_probe.py glove 100,200,400,800 occupancy_target=$ot 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/ot$ot /" | tee -a $O/occ_ot.txt
done
