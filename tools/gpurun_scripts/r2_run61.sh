#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run61; mkdir -p $O
for kind in sift glove sift_u8 s3; do
  timeout 900 python tools/occ_probe.py $kind 16,32,52,64,100,128,200 2>&1 | grep -v amdgpu | tee -a $O/occ.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
