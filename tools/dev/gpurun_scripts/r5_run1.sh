#!/bin/bash
# round 5, run 1: the whole GPU suite on the new tree, then the default bench invocation (contract line must parse, <= 4 KB)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run1
mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --full-record $O/bench_full.json > $O/bench_line.json 2> $O/bench.log; echo "bench rc=$?"
wc -c $O/bench_line.json
cat $O/bench_line.json
grep "^\[bench\] c" $O/bench.log
