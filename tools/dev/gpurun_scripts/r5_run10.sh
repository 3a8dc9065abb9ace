#!/bin/bash
# round 5, run 10: 10M x 768 at ef=670 -- what does fnv_tune measure for the variants (its choice flipped between two boxes:
# merged-beam kernel alone 97.7 ms vs 75 % exact tail 107.4 ms), pinned variants A/B
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run10
mkdir -p $O
cd $R
FLATNAV_TUNE_LOG=1 timeout 1500 python tools/dev/knob_sweep.py --config c3-lowrank --ef 670 --rounds 2 --steps 6 --nb 6 --sets \
  "base" "sorted_variant=1" "sorted_variant=3" "sorted_variant=5" "sorted_variant=2" > $O/sweep_c3lowrank.txt 2>&1; echo "rc=$?"
grep "fnv_tune B=670 variant\|fnv_tune B=670 layout" $O/sweep_c3lowrank.txt | head -40
grep -v "^\[\|amdgpu.ids\|fnv_tune" $O/sweep_c3lowrank.txt | tail -14
