#!/bin/bash
# round 3, run 47: 20 resident float32 queries per CU (96 VGPRs per lane, 2048-slot table, exact search's heap in HBM) vs the default 16
mkdir -p gpurun_out/r3_run47
O=gpurun_out/r3_run47
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # lib, tag, args
  FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$1.so timeout 600 python bench.py $QUICK $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 | lib$1', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['launch']['lds_bytes'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
line "" "c2 default" ""
line "_w5" "c2 w5 default-layout" ""
line "_w5" "c2 w5 2048+heapHBM" "--opt visited_slots=2048 --opt sorted_cand_lds=0"
line "_w5" "c2 w5 1536+heapHBM" "--opt visited_slots=1536 --opt sorted_cand_lds=0"
line "" "c2 w4 2048+heapHBM" "--opt visited_slots=2048 --opt sorted_cand_lds=0"
line "_w5" "c2 w5 2048+heapHBM tail0" "--opt visited_slots=2048 --opt sorted_cand_lds=0 --opt sorted_variant=1"
cat $O/lines.txt
