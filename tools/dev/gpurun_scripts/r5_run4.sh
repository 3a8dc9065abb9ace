#!/bin/bash
# round 5, run 4: the whole GPU suite on the hand-over tree; small-batch latency with shadows vs replay
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n 'FAILED\|passed\|failed' $O/pytest.log | tail -20
timeout 900 python tools/dev/latency_probe.py 1000000 > $O/latency_shadow.txt 2>&1; echo "rc=$?"
timeout 900 python tools/dev/latency_probe.py 1000000 shadow_exact=0 > $O/latency_replay.txt 2>&1; echo "rc=$?"
grep -v "caller threads" $O/latency_shadow.txt | head -30; echo ---; grep -v "caller threads" $O/latency_replay.txt | head -30
