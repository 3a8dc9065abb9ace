#!/bin/bash
# round 4, run 21: the guessed next node's unvisited neighbour VECTORS requested one hop ahead (-DFNV_EXP_PREFETCH_VEC build
# = pf) against the tree's library, alternating on one copy of each index
O=gpurun_out/r4_run21; mkdir -p $O
E=flatnav_amd/_exp
timeout 300 python tools/dev/knob_sweep.py --config c4 --ef 110,200 --rounds 3 --steps 10 --libs pf=$E/libflatnav_hip_pf.so --sets base pf:base > $O/c4.txt 2>$O/c4.err; cat $O/c4.txt
timeout 300 python tools/dev/knob_sweep.py --config c2 --ef 52,100 --rounds 3 --steps 10 --libs pf=$E/libflatnav_hip_pf.so --sets base pf:base > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt
timeout 600 python tools/dev/knob_sweep.py --config c5-lowrank --n 20000000 --ef 76 --rounds 3 --steps 8 --libs pf=$E/libflatnav_hip_pf.so --sets base pf:base > $O/c5.txt 2>$O/c5.err; cat $O/c5.txt; tail -3 $O/c5.err
