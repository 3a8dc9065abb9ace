#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run30; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for kind in glove sift; do
    timeout 600 python tools/occ_probe.py $kind 52,100,200,400,800 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
done
timeout 900 python tools/occ_probe.py s3 200,800 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
