#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run71; mkdir -p $O
for spec in "sorted_cand_lds=2" "sorted_cand_lds=1 visited_slots=3072" "sorted_cand_lds=1 visited_slots=4096" "sorted_cand_lds=1"; do
  timeout 600 python tools/occ_probe.py sift_u8 32,52,64,100,128 $spec sorted_tail_exact_pct=50 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/[$spec] /" | tee -a $O/occ.txt
done
