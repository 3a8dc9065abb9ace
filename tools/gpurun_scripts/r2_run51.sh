#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run51; mkdir -p $O
timeout 600 python tools/occ_probe.py sift_u8 52,100,200,400 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/u8.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "uint8 or integer_dtypes or random_shapes" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
