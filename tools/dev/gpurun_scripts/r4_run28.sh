#!/bin/bash
# round 4, run 28: 10M x 768 at ef=670 -- the right-sized grid (10 really resident slots per CU) against the grid the occupancy
# API's count gives (11 per CU, 256 workgroups of it never resident until others exit): the bench lost 8 % with the former
O=gpurun_out/r4_run28; mkdir -p $O
E=flatnav_amd/_exp
timeout 800 python tools/dev/knob_sweep.py --config c3-lowrank --ef 670 --rounds 3 --steps 4 --libs old=$E/libflatnav_hip_old.so \
  --sets base old:base sorted_variant=3 sorted_variant=4 sorted_variant=2 old:sorted_variant=3 > $O/c3.txt 2>$O/c3.err; cat $O/c3.txt; tail -2 $O/c3.err
