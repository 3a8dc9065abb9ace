#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run13; mkdir -p $O
timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift.txt 2>&1; grep -v amdgpu $O/time_sift.txt | tail -30
timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove.txt 2>&1; grep -v amdgpu $O/time_glove.txt | tail -30
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
