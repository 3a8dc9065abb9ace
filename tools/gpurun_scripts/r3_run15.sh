#!/bin/bash
# round 3, run 15: fewer resident queries -> shorter per-query latency under load -> shorter drain?  (c2 float32, c4 ef=110)
mkdir -p gpurun_out/r3_run15
O=gpurun_out/r3_run15
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 5"
for b in 16 14 12 10 8; do
  for a in "--dtype float32 --ef 52" "--config c4 --ef 110"; do
  python bench.py $QUICK $a --opt blocks_per_cu=$b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bpc=$b $a', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
  done
done
cat $O/lines.txt
