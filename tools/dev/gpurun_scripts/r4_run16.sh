#!/bin/bash
# round 4, run 16: the committed tree as the driver runs it -- the whole GPU suite, smoke(), the default bench invocation
O=gpurun_out/r4_run16; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time python bench.py --gpus 1 ) > $O/bench.json 2> $O/bench.err
grep "^\[bench\]" $O/bench.err | tail -7; grep "host-buffer" $O/bench.err
tail -3 $O/bench.err
