#!/bin/bash
# round 3, run 28: per-phase cycles of one query alone / 64 / 10000, with and without the runner-up's link row one hop ahead
mkdir -p gpurun_out/r3_run28
O=gpurun_out/r3_run28
timeout 600 python tools/phase_profile.py --ef 100 > $O/spec.txt 2>&1
timeout 600 python tools/phase_profile.py --ef 100 --tag nospec > $O/nospec.txt 2>&1
grep -A14 "sorted beam" $O/spec.txt; echo ======; grep -A14 "sorted beam" $O/nospec.txt
