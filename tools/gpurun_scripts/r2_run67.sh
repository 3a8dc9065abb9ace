#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run67; mkdir -p $O
for spec in "visited_slots=3072 sorted_cand_lds=0" "visited_slots=4096 sorted_cand_lds=0" "visited_slots=2048 sorted_cand_lds=0" "visited_slots=6144 sorted_cand_lds=0 blocks_per_cu=16" "visited_slots=3072 sorted_cand_lds=1" "visited_slots=4096 sorted_cand_lds=1"; do
  timeout 600 python tools/occ_probe.py glove 52,100 $spec 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/[$spec] /" | tee -a $O/occ.txt
done
