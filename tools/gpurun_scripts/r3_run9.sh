#!/bin/bash
# round 3, run 9: the exact search's fused pushes / top from registers: parity (all exact-kernel suites) + bench lines
mkdir -p gpurun_out/r3_run9
O=gpurun_out/r3_run9
FNV_FULLSIZE=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 5"
for a in "--dtype float32" "--dtype uint8" "--dtype float32 --ef 100" "--dtype uint8 --ef 100" "--dtype float32 --opt sorted_beam=0" "--dtype uint8 --opt sorted_beam=0" "--index-size 1000000 --config c5-lowrank --ef 80"; do
  python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'], d['config']['queries_replayed_by_exact_kernel'])" >> $O/bench_lines.txt 2>&1
done
cat $O/bench_lines.txt
python tools/latency_probe.py > $O/latency.txt 2>&1; tail -12 $O/latency.txt
