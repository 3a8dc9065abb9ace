#!/bin/bash
# round 3, run 44: the uint8 index with the stash (prev) and without it (1-byte rows skip it), alternating; uint8 parity tests first
mkdir -p gpurun_out/r3_run44
O=gpurun_out/r3_run44
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uint8 or u8 or integer or wide or spill or ties" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # lib, tag, args
  FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$1.so timeout 600 python bench.py $QUICK $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 | lib$1', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
for rep in 1 2; do
  for lib in _prev ""; do
    line "$lib" "u8" "--dtype uint8"
    line "$lib" "u8-100" "--dtype uint8 --ef 100"
  done
done
sort -s -k1,1 $O/lines.txt
