#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run22; mkdir -p $O
for mb in 0 2; do
  timeout 600 python tools/latency_probe.py 1000000 merged_beam=$mb 2>&1 | grep -v amdgpu | grep -v "batch  *[0-9]*[46]:" | sed "s/^/mb$mb /" | tee -a $O/latency.txt
done
for mb in 0 2; do
  timeout 600 python tools/occ_probe.py sift_u8 32,52,64,100,200 merged_beam=$mb 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/mb$mb /" | tee -a $O/occ_u8.txt
done
for vs in 0 8192 16384; do
  timeout 600 python tools/occ_probe.py glove 200,400,800 visited_slots=$vs 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/vs$vs /" | tee -a $O/occ_vis.txt
done
