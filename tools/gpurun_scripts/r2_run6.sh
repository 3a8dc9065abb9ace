#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run6; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for cfg in "c2" "c2 --dtype uint8" "c4"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print(round(d["value"]), "ef", d["config"]["ef_search"], "recall", d["config"]["recall_at_10"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"]["kernel"], "reruns", d["config"]["queries_replayed_by_exact_kernel"], "sustained", round(d["sustained"]["value"]))
for s in d["secondary"]: print("   ef", s["ef_search"], round(s["value"]), s["recall_at_10"], round(s["roofline_frac"],3))
print("   cpu", round(d["cpu_baseline"]["value"]), d["config"]["index_build"])
PY
done
