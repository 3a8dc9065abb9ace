#!/bin/bash
# round 3, run 25: run-to-run variance of the C3-shaped line on one box (same library, fresh process each time)
mkdir -p gpurun_out/r3_run25
O=gpurun_out/r3_run25
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 --warmup 3 --index-size 3000000 --config c3-lowrank --ef 800"
for rep in 1 2 3 4 5; do
  rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "Temperature \(Sensor (edge|junction)|sclk|Average Graphics Package Power|Current Socket" | head -4 | tr '\n' ' ' >> $O/lines.txt
  python bench.py $QUICK 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' | rep $rep', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['gather_ceiling']), d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
done
cat $O/lines.txt
