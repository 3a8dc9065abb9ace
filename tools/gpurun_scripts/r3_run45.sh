#!/bin/bash
# round 3, run 45: smoke(), then fnv_tune timing from cold caches (default) vs warm (FLATNAV_TUNE_FLUSH=0): which variant / layout it settles on
# and what the bench protocol then measures -- uint8 index (256 MB: fits the Infinity Cache) and c2 float32 (640 MB)
mkdir -p gpurun_out/r3_run45
O=gpurun_out/r3_run45
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # flush, tag, args
  FLATNAV_TUNE_FLUSH=$1 timeout 600 python bench.py $QUICK $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 | flush=$1', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
for rep in 1 2 3; do
  for fl in 1 0; do
    line $fl "u8" "--dtype uint8"
  done
done
for rep in 1 2; do
  for fl in 1 0; do
    line $fl "c2" ""
    line $fl "c4-110" "--config c4 --ef 110"
  done
done
sort -s -k1,1 $O/lines.txt
