#!/bin/bash
# round 3, run 19: wide-tag visited probe on 32-bit halves: parity (forced wide tags, 50M-node full-size test) + c5 / c5-lowrank lines
mkdir -p gpurun_out/r3_run19
O=gpurun_out/r3_run19
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python tools/bigid_check.py > $O/bigid.txt 2>&1; tail -3 $O/bigid.txt
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 --warmup 3"
for a in "--config c5" "--config c5-lowrank"; do
  python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), d['config']['ef_search'], d['config']['recall_at_10'], d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'])" >> $O/lines.txt 2>&1
done
cat $O/lines.txt
