#!/bin/bash
# round 3, run 11: more vectors in flight for rows of 3 KB and more (768-d float32): 4 (shipped) vs 6 vs 8 passes, 2 waves per SIMD
mkdir -p gpurun_out/r3_run11
O=gpurun_out/r3_run11
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 --warmup 3 --index-size 3000000 --config c3-lowrank"
for lib in "" _pu6 _pu8; do
  for ef in 200 800; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$lib.so python bench.py $QUICK --ef $ef 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib$lib ef $ef', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), round(d['roofline']['gather_ceiling']), d['config']['launch'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
  done
done
cat $O/lines.txt
