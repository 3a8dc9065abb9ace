#!/bin/bash
# round 4, run 23: c2 at ef=52 -- resident slots per CU against the launch shape (2.44 rounds of 4096 slots at 16 per CU)
O=gpurun_out/r4_run23; mkdir -p $O
timeout 500 python tools/dev/knob_sweep.py --config c2 --ef 52 --rounds 3 --steps 10 \
  --sets base blocks_per_cu=10 blocks_per_cu=12 blocks_per_cu=13 blocks_per_cu=14 blocks_per_cu=15 > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt; tail -3 $O/c2.err
