#!/bin/bash
# round 3, run 35: what fnv_tune measures for c2 ef=200 inside bench.py and inside tools/layout_ab.py
mkdir -p gpurun_out/r3_run35
O=gpurun_out/r3_run35
export FLATNAV_TUNE_LOG=1
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
timeout 600 python bench.py $QUICK --ef 200 2>$O/bench_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch'], d['config']['kernel_variant'])" > $O/lines.txt
grep fnv_tune $O/bench_err.txt >> $O/lines.txt
timeout 900 python tools/layout_ab.py c2 200 >> $O/lines.txt 2>$O/ab_err.txt
grep fnv_tune $O/ab_err.txt | head -40 >> $O/lines.txt
cat $O/lines.txt
