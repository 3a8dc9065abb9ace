#!/bin/bash
# round 3, run 24: early vs late link-row request on the C3 shape (768-d, 3M nodes), same box, alternating
mkdir -p gpurun_out/r3_run24
O=gpurun_out/r3_run24
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 --warmup 3 --index-size 3000000 --config c3-lowrank"
for rep in 1 2; do
for lib in "" _late; do
  for ef in 200 800; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$lib.so python bench.py $QUICK --ef $ef 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib$lib ef $ef', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac_of_gather_ceiling'],3), round(d['roofline']['gather_ceiling']), d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
  done
done
done
sort $O/lines.txt
