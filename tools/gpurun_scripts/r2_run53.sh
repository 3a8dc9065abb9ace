#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run53; mkdir -p $O
for cfg in "c3-lowrank" "c3"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 1500 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print(round(d["value"]), "ef", d["config"]["ef_search"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"]["kernel"], d["config"]["launch"]["tail_exact"], "reruns", d["config"]["queries_replayed_by_exact_kernel"], d["config"]["index_build"][-8:])
for s in d["secondary"]: print("   ef", s["ef_search"], round(s["value"]), round(s["roofline_frac"],3))
PY
done
