#!/bin/bash
# round 4, run 30 (the round's last GPU seconds): c2 at ef=52 with a grid of ceil(nq / rounds) slots -- every slot exactly three queries
O=gpurun_out/r4_run30; mkdir -p $O
timeout 100 python tools/dev/knob_sweep.py --config c2 --ef 52 --rounds 2 --steps 10 --sets base even_rounds=1 grid_slots=3400 grid_slots=3584 > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt; tail -2 $O/c2.err
