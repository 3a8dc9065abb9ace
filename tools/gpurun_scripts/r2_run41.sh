#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run41; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for kind in sift sift_u8; do
    timeout 600 python tools/occ_probe.py $kind 32,52,100,200,400 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
done
