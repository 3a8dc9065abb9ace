#!/bin/bash
# round 4, run 24: the shape of a launch -- busy slots and finished queries per slice of its duration (a -DFNV_TIMELINE build)
O=gpurun_out/r4_run24; mkdir -p $O
L=tl=flatnav_amd/_exp/libflatnav_hip_tl.so
timeout 300 python tools/dev/launch_timeline.py --config c2 --ef 52 --lib $L --variants=-1,1 > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt; tail -3 $O/c2.err
timeout 300 python tools/dev/launch_timeline.py --config c2 --ef 52 --dtype uint8 --lib $L --variants=-1,1 > $O/c2u8.txt 2>$O/c2u8.err; tail -3 $O/c2u8.err
timeout 500 python tools/dev/launch_timeline.py --config c5-lowrank --n 20000000 --ef 76 --lib $L --variants=-1,1 > $O/c5.txt 2>$O/c5.err; tail -3 $O/c5.err
