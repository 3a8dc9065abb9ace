#!/bin/bash
# round 5, run 11: 10M x 768 at ef=670 under the bench protocol: the rules' layout (4096-slot table, 11 per CU) against the
# 6144-slot table (8 per CU) that fnv_tune measures 4.4 % faster but rejects (5 % margin)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run11
mkdir -p $O
cd $R
timeout 1500 python tools/dev/knob_sweep.py --config c3-lowrank --ef 670 --rounds 3 --steps 8 --nb 8 --sets \
  "base" "visited_slots=6144" "visited_slots=6144,sorted_variant=1" "sorted_variant=1" "visited_slots=8192" > $O/sweep_c3lowrank.txt 2>&1; echo "rc=$?"
grep -v "^\[\|amdgpu.ids\|fnv_tune" $O/sweep_c3lowrank.txt | tail -14
