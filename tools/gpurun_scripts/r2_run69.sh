#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run69; mkdir -p $O
for kind in glove sift sift_u8 s3; do
  timeout 900 python tools/occ_probe.py $kind 52,64,100,128,200 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
done
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
