#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run17; mkdir -p $O
timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift.txt 2>&1; grep -v amdgpu $O/time_sift.txt | cut -c1-100 | tail -50
timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove.txt 2>&1; grep -v amdgpu $O/time_glove.txt | cut -c1-100 | tail -50
