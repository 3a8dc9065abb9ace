#!/bin/bash
# round 5, run 14: where does a loaded hop's time go on 20M x 128 (the 50M configuration's generator) -- per-phase shader
# cycles (a -DFNV_PHASE_TIMING build on the tree's index) at the rules' table, a bigger and a smaller one
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run14
mkdir -p $O
cd $R
timeout 1500 python tools/dev/knob_sweep.py --config c5-lowrank --n 20000000 --ef 80 --rounds 1 --steps 6 --nb 6 --libs prof=flatnav_amd/_exp/libflatnav_hip_prof.so --sets \
  "base" "prof:sorted_variant=1" "prof:sorted_variant=1,visited_slots=6144" "prof:sorted_variant=1,visited_slots=2048" "sorted_variant=1" > $O/phases_c5lowrank_20m.txt 2>&1; echo "rc=$?"
grep -v "^\[\|amdgpu.ids" $O/phases_c5lowrank_20m.txt | tail -16
timeout 900 python tools/dev/knob_sweep.py --config c5 --n 20000000 --ef 100 --rounds 1 --steps 6 --nb 6 --libs prof=flatnav_amd/_exp/libflatnav_hip_prof.so --sets \
  "base" "prof:sorted_variant=1" > $O/phases_c5_20m.txt 2>&1; echo "rc=$?"
grep -v "^\[\|amdgpu.ids" $O/phases_c5_20m.txt | tail -8
