#!/bin/bash
# round 3, run 46: resident queries per CU vs the drain -- does a bandwidth-bound launch finish sooner with fewer, faster queries in flight?
mkdir -p gpurun_out/r3_run46
O=gpurun_out/r3_run46
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # tag, args
  timeout 600 python bench.py $QUICK $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
for b in 16 14 12 10 8; do line "c2 bpc=$b" "--opt blocks_per_cu=$b"; done
for b in 12 10 8; do line "c2-ef100 bpc=$b" "--ef 100 --opt blocks_per_cu=$b"; done
cat $O/lines.txt
