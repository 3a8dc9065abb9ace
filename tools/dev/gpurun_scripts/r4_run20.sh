#!/bin/bash
# round 4, run 20: one big host call, plain vs sharded over the handle's own lanes
O=gpurun_out/r4_run20; mkdir -p $O
timeout 300 python tools/dev/big_batch_bench.py > $O/big.txt 2> $O/big.err; cat $O/big.txt; tail -3 $O/big.err
