#!/bin/bash
# round 3, run 12: bitmap words requested early (visited-set overflow no longer a third dependent round trip): parity + wide beams
mkdir -p gpurun_out/r3_run12
O=gpurun_out/r3_run12
FNV_FULLSIZE=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 --warmup 3"
for a in "--config c4 --ef 110" "--config c4 --ef 200" "--config c4 --ef 400" "--dtype float32 --ef 200" "--dtype float32 --ef 400" "--dtype uint8 --ef 400" "--index-size 3000000 --config c3-lowrank --ef 200" "--index-size 3000000 --config c3-lowrank --ef 800" "--dtype float32" "--dtype uint8"; do
  python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
done
cat $O/lines.txt
