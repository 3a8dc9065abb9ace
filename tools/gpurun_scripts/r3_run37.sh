#!/bin/bash
# round 3, run 37: whole GPU suite + the default bench line with the overflow stash, the link-row guess and the settled tuner
mkdir -p gpurun_out/r3_run37
O=gpurun_out/r3_run37
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 1500 python bench.py > $O/bench.json 2>$O/bench_err.txt
echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3_run37/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_gather_ceiling"), d["config"].get("kernel_variant"), d["config"]["launch"])
for k,v in d.get("secondary",{}).items():
    if isinstance(v,dict) and "value" in v:
        print(k, v["value"], v.get("recall"), v["roofline"]["frac"], v["roofline"].get("frac_of_gather_ceiling"), v["config"].get("ef_search"), v["config"].get("kernel_variant"))
        for p in v.get("sweep",[]) or []: print("   ", p)
P
