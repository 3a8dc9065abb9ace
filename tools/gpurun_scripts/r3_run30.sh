#!/bin/bash
# round 3, run 30: per-phase cycles of one query alone (no shadow), with and without the runner-up's link row one hop ahead
mkdir -p gpurun_out/r3_run30
O=gpurun_out/r3_run30
timeout 600 python tools/phase_profile.py --ef 100 --opt shadow_exact=0 > $O/spec.txt 2>&1
timeout 600 python tools/phase_profile.py --ef 100 --opt shadow_exact=0 --tag nospec > $O/nospec.txt 2>&1
grep -A12 "sorted beam, 1 q" $O/spec.txt; echo ======; grep -A12 "sorted beam, 1 q" $O/nospec.txt
grep -A12 "sorted beam, 10000 q" $O/spec.txt; echo ======; grep -A12 "sorted beam, 10000 q" $O/nospec.txt
