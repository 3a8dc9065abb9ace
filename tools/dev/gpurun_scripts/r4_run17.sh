#!/bin/bash
# round 4, run 17: last check of the committed tree -- build() as the driver runs it, the GPU suite, smoke(), a c2 + c4 bench line
O=gpurun_out/r4_run17; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --secondary-configs c4 --steps 20 > $O/bench.json 2> $O/bench.err
grep "^\[bench\]\|host-buffer" $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['summary'], len(json.dumps(d)))"
