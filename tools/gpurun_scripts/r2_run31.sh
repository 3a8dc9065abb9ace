#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run31; mkdir -p $O
for ef in 100 400; do
timeout 600 python tools/phase_profile.py --ef $ef --nq 10000 --opt sorted_tail_exact_pct=0 2>&1 | grep -v amdgpu > $O/phase_ef$ef.txt
done
grep -A14 "sorted beam, 10000 queries" $O/phase_ef*.txt
grep -A14 "sorted beam, 1 queries" $O/phase_ef400.txt
