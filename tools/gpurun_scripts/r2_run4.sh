#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run4; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift.txt 2>&1; tail -30 $O/time_sift.txt
timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove.txt 2>&1; tail -30 $O/time_glove.txt
