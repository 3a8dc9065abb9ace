#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run64; mkdir -p $O
for vs in 6144; do
  timeout 1500 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --sustain-seconds 0 --opt visited_slots=$vs > $O/bench_c5_vs$vs.json 2> $O/bench_c5_vs$vs.log; echo "rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_c5_vs$vs.json"))
print("vs$vs", round(d["value"]), "frac", round(d["roofline"]["frac"],3), d["config"]["launch"])
PY
done
