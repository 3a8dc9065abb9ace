#!/bin/bash
# round 3, run 43: the GPU suite with per-test durations (run 42 took 22 min instead of 5: which tests, or which box?)
mkdir -p gpurun_out/r3_run43
O=gpurun_out/r3_run43
nproc > $O/host.txt; uptime >> $O/host.txt
timeout 1150 python -m pytest tests -m gpu -x -q --durations=30 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
uptime >> $O/host.txt
tail -45 $O/pytest.log; cat $O/host.txt
