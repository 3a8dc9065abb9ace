#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run66; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "c5" "c3" "c3-lowrank"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 1500 python bench.py --config $cfg --steps 10 --warmup 3 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print(round(d["value"]), "ef", d["config"]["ef_search"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"], "cpu", round(d["cpu_baseline"]["value"]))
for s in d["secondary"]: print("   ef", s["ef_search"], round(s["value"]), round(s["roofline_frac"],3))
PY
done
