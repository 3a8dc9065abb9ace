#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run59; mkdir -p $O
for kind in glove sift sift_u8 s3; do
  timeout 900 python tools/occ_probe.py $kind 100,128,160,200,256,400 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
