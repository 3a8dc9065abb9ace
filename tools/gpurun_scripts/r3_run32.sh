#!/bin/bash
# round 3, run 32: stash with retirement -- parity, then wide beams and the mid beams again, base vs new
mkdir -p gpurun_out/r3_run32
O=gpurun_out/r3_run32
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 3"
line() {  # lib, tag, args
  FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$1.so timeout 600 python bench.py $QUICK $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 | lib$1', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac_of_gather_ceiling'],3), d['config']['launch']['blocks_per_cu'], d['config']['launch']['visited_slots'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
}
for rep in 1 2; do
  for lib in _base ""; do
    line "$lib" "c4-400" "--config c4 --ef 400"
    line "$lib" "c2-400" "--ef 400"
    line "$lib" "c2-200" "--ef 200"
    line "$lib" "c4-200" "--config c4 --ef 200"
    line "$lib" "c4-110" "--config c4 --ef 110"
    line "$lib" "c2-ef100" "--ef 100"
    line "$lib" "u8" "--dtype uint8"
    line "$lib" "u8-100" "--dtype uint8 --ef 100"
  done
done
sort -s -k1,1 $O/lines.txt
