#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run2; mkdir -p $O
timeout 600 python tools/r2_probe.py reasons > $O/reasons.txt 2>&1; cat $O/reasons.txt | tail -15
timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift.txt 2>&1; tail -20 $O/time_sift.txt
timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove.txt 2>&1; tail -20 $O/time_glove.txt
timeout 1500 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_configs.py tests/test_gpu_python_api.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
