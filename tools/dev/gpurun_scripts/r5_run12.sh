#!/bin/bash
# round 5, run 12: the tree as the driver will run it: build(), whole GPU suite, smoke(), default bench invocation
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run12
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n 'FAILED\|passed\|failed' $O/pytest.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
python bench.py --steps 20 --warmup 5 --full-record $O/bench_full.json > $O/bench_line.json 2> $O/bench.log; echo "bench rc=$?"
wc -c $O/bench_line.json; cat $O/bench_line.json
grep "^\[bench\] c" $O/bench.log | tail -7
