#!/bin/bash
# round 3, run 39: c5 (50M x 128 randn): base vs current without the stash (compile-time) vs current
mkdir -p gpurun_out/r3_run39
O=gpurun_out/r3_run39
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # lib, tag, args
  FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$1.so timeout 900 python bench.py $QUICK $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 | lib$1', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac_of_gather_ceiling'],3), d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'], d['config'].get('mean_dist_evals_per_query'))" >> $O/lines.txt 2>&1
}
for lib in _nostash "" _base; do
  line "$lib" "c5" "--config c5 --ef 100"
done
sort -s -k1,1 $O/lines.txt
