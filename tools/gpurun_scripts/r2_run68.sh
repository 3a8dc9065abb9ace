#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run68; mkdir -p $O
for spec in "sorted_cand_lds=2" "sorted_cand_lds=1" "sorted_cand_lds=1 visited_slots=4096" "sorted_cand_lds=1 visited_slots=6144"; do
  for kind in glove sift s3; do
  timeout 900 python tools/occ_probe.py $kind 64,100,128,200,256,400 $spec 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/[$spec] /" | tee -a $O/occ.txt
  done
done
