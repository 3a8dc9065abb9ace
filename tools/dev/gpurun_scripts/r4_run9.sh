#!/bin/bash
# round 4, run 9: the whole GPU suite, then the default bench invocation (seven configurations), as the driver runs them
O=gpurun_out/r4_run9; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
( time python bench.py --gpus 1 ) > $O/bench.json 2> $O/bench.err
tail -12 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_run9/bench.json").read().strip().splitlines()[-1])
print("\n".join(d["summary"])); print(d["bench_wall_seconds"], d["config"]["host_buffer_qps_pcie_inclusive"])
PY
