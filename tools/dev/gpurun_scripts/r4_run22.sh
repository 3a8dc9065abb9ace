#!/bin/bash
# round 4, run 22: where the wide beams of the 1.18M x 100 index (c4, ef = 200 / 400: the reference's fixed-ef lines) lose --
# visited-table sizes, the overflow list, per-phase cycles
O=gpurun_out/r4_run22; mkdir -p $O
E=flatnav_amd/_exp
timeout 500 python tools/dev/knob_sweep.py --config c4 --ef 400,200 --rounds 2 --steps 8 --libs prof=$E/libflatnav_hip_prof.so \
  --sets base visited_slots=8192 visited_slots=12288 visited_slots=16384 sorted_variant=1 prof:sorted_variant=1 prof:visited_slots=16384,sorted_variant=1 > $O/c4.txt 2>$O/c4.err; cat $O/c4.txt; tail -3 $O/c4.err
