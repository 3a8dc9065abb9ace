#!/bin/bash
# round 3, run 17: the link row requested before the tie bookkeeping of the selection (early) vs after it (late)
mkdir -p gpurun_out/r3_run17
O=gpurun_out/r3_run17
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 5"
for rep in 1 2; do
for lib in "" _late; do
  for a in "--dtype float32 --ef 52" "--dtype uint8 --ef 52" "--config c4 --ef 110" "--config c4 --ef 200" "--config c4 --ef 400" "--index-size 1000000 --config c5-lowrank --ef 80"; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$lib.so python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib$lib $a', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'], round(d['pipelined']['value']) if d.get('pipelined') else '')" >> $O/lines.txt 2>&1
  done
done
done
sort $O/lines.txt
