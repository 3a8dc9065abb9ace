#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run23; mkdir -p $O
for ef in 52 100 200; do
timeout 600 python tools/phase_profile.py --ef $ef --nq 64 --opt merged_beam=2 2>&1 | grep -v amdgpu > $O/phase_ef$ef.txt
done
grep -A12 "sorted beam, 1 queries" $O/phase_ef*.txt
