#!/bin/bash
# round 4, run 1: knob sweep with the round-3 library -- what do residency / visited-table size buy on the 30 GB tables?
O=gpurun_out/r4_run1; mkdir -p $O
timeout 1100 python tools/dev/knob_sweep.py --config c3-lowrank --ef 650,700,750,800 --recall --rounds 2 --steps 5 \
  --sets base visited_slots=1536 visited_slots=768 visited_slots=3072 > $O/c3.txt 2>$O/c3.err
tail -20 $O/c3.txt
timeout 700 python tools/dev/knob_sweep.py --config c5-lowrank --ef 72,76,80 --recall --rounds 2 --steps 8 \
  --sets base visited_slots=1536 visited_slots=3072 visited_slots=1536,sorted_cand_lds=0 visited_slots=3072,sorted_cand_lds=0 visited_slots=6144,sorted_cand_lds=0 > $O/c5l.txt 2>$O/c5l.err
tail -20 $O/c5l.txt
