#!/bin/bash
# round 5, run 8: two launches in flight, tree vs the round-4 library (bench reported 9.7 M on c2 where round 4 had 12.0 M)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run8
mkdir -p $O
cd $R
timeout 600 python tools/dev/pipelined_probe.py --config c2 --lib r4=flatnav_amd/_exp/libflatnav_hip_r4.so > $O/pipelined_c2.txt 2>&1; grep -v amdgpu.ids $O/pipelined_c2.txt | tail -6
timeout 600 python tools/dev/pipelined_probe.py --config c2 --dtype uint8 --lib r4=flatnav_amd/_exp/libflatnav_hip_r4.so > $O/pipelined_c2_uint8.txt 2>&1; grep -v amdgpu.ids $O/pipelined_c2_uint8.txt | tail -6
