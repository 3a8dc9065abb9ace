#!/bin/bash
# round 3, run 34: bench lines with fnv_tune's layout step on / off (c2 ef=200, c4 ef=400)
mkdir -p gpurun_out/r3_run34
O=gpurun_out/r3_run34
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # tag, args
  timeout 600 python bench.py $QUICK $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac_of_gather_ceiling'],3), d['config']['launch'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
for rep in 1 2; do
line "c2-200 tuned" "--ef 200"
line "c2-200 rules" "--ef 200 --opt tune_layout=0"
line "c4-400 tuned" "--config c4 --ef 400"
line "c4-400 rules" "--config c4 --ef 400 --opt tune_layout=0"
done
sort -s -k1,2 $O/lines.txt
