#!/bin/bash
# round 3, run 5: layout tuning inside fnv_tune + the 25 % tail: parity, then every bench line quickly
mkdir -p gpurun_out/r3_run5
O=gpurun_out/r3_run5
FNV_FULLSIZE=0 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "rules|passed|failed|rc=" $O/pytest.log | tail -12
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 5"
for a in "--dtype float32" "--dtype uint8" "--dtype float32 --ef 100" "--dtype uint8 --ef 100" "--config c4 --ef 50" "--config c4 --ef 110" "--config c4 --ef 200" "--config c4 --ef 400" "--index-size 1000000 --config c5-lowrank --ef 80" "--index-size 2000000 --config c3-lowrank --ef 200" "--index-size 2000000 --config c3-lowrank --ef 400"; do
  python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'], d['config']['queries_replayed_by_exact_kernel'], d['config']['kernel_choice'][-70:])" >> $O/bench_lines.txt 2>&1
done
cat $O/bench_lines.txt
