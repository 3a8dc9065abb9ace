#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run3; mkdir -p $O
python -m flatnav_amd.build > $O/build.log 2>&1
timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift.txt 2>&1; tail -30 $O/time_sift.txt
timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove.txt 2>&1; tail -30 $O/time_glove.txt
