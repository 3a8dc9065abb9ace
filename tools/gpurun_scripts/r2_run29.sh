#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run29; mkdir -p $O
for mb in 0 1; do
  timeout 900 python tools/occ_probe.py s3 100,200,400,800 merged_beam=$mb 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/mb$mb /" | tee -a $O/occ.txt
done
