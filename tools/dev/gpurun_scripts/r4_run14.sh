#!/bin/bash
# round 4, run 14: the second lane for concurrent host-buffer callers -- tests, then the c2 line's host-buffer rates
O=gpurun_out/r4_run14; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_python_api.py tests/test_gpu_multi_device.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for c in c2 c2-uint8 c4; do
  timeout 300 python bench.py --config $c --secondary-configs none --no-cpu-baseline --steps 20 > $O/bench_$c.json 2> $O/bench_$c.err
  grep "host-buffer" $O/bench_$c.err
  python - <<PY
import json
d = json.load(open("$O/bench_$c.json"))
print("$c", round(d["value"]), d["config"]["host_buffer_qps_pcie_inclusive"], d["config"]["host_buffer_qps_two_caller_threads"], round(d["pipelined"]["value"]))
PY
done
