#!/bin/bash
# round 3, run 41: whole GPU suite, then the round's profile set (tools/collect_profiles.sh) with the stash library
mkdir -p gpurun_out/r3_run41
O=gpurun_out/r3_run41
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 2400 bash tools/collect_profiles.sh > $O/collect.log 2>&1
echo "collect rc=$?"
tail -5 $O/collect.log
