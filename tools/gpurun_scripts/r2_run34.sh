#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run34; mkdir -p $O
for ef in 52 100 400; do
timeout 600 python tools/phase_profile.py --ef $ef --nq 10000 --opt sorted_tail_exact_pct=0 2>&1 | grep -v amdgpu > $O/phase_ef$ef.txt
done
timeout 600 python tools/latency_probe.py 1000000 2>&1 | grep -v amdgpu > $O/latency.txt
tail -3 $O/phase_ef52.txt; tail -12 $O/latency.txt
