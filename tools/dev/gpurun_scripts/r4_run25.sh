#!/bin/bash
# round 4, run 25: after the launch timelines (run 24: a launch is whole ROUNDS of its slots -- 10 000 queries on 4096 slots are
# three rounds, the third 44 % full) and the LDS-granule probe (gfx950 allocates LDS in 1280-byte granules: 7712 bytes per
# slot are 18 per CU, not the 21 the occupancy API reports): MORE SLOTS so that 10 000 queries are two rounds.
#   uint8: 32 bytes less LDS (cand_slots 296 -> 292: 7680 bytes = 6 granules) -> 21 per CU really resident
#   float32: the one-chunk merged-beam kernel compiled for five waves per SIMD (<= 96 registers: two passes of vectors in
#   flight without spills = w5p2, three with 68 bytes of scratch = w5p3) + 7680 bytes of LDS (2048-slot table, 243-entry heap)
O=gpurun_out/r4_run25; mkdir -p $O
E=flatnav_amd/_exp
timeout 400 python tools/dev/knob_sweep.py --config c2 --dtype uint8 --ef 52 --rounds 3 --steps 10 \
  --sets base cand_slots=292 visited_slots=2048,cand_slots=292 visited_slots=2048 > $O/c2u8.txt 2>$O/c2u8.err; cat $O/c2u8.txt; tail -2 $O/c2u8.err
timeout 500 python tools/dev/knob_sweep.py --config c2 --ef 52,100 --rounds 3 --steps 10 --libs w5p2=$E/libflatnav_hip_w5p2.so,w5p3=$E/libflatnav_hip_w5p3.so \
  --sets base visited_slots=2048,cand_slots=243 w5p2:base w5p2:visited_slots=2048,cand_slots=243 w5p3:visited_slots=2048,cand_slots=243 w5p2:visited_slots=2048,sorted_cand_lds=0 w5p2:visited_slots=1536,cand_slots=243 > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt; tail -2 $O/c2.err
