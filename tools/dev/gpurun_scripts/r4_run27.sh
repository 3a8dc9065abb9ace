#!/bin/bash
# round 4, run 27: the whole GPU suite, smoke() and the default bench invocation on the granule-aware tree
O=gpurun_out/r4_run27; mkdir -p $O
timeout 700 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1100 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep "^\[bench\]" $O/bench.log | tail -12
