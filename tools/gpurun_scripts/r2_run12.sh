#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_bench.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for cfg in "c2" "c2 --dtype uint8" "c4"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print(round(d["value"]), "ef", d["config"]["ef_search"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"]["kernel"], "sustained", round(d["sustained"]["value"]), "PIPELINED", round(d["pipelined"]["value"]), d["pipelined"]["steps"])
PY
done
