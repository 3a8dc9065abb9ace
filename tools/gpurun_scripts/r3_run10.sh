#!/bin/bash
# round 3, run 10: shadow mode (exact shadows for small launches): parity + small-batch latency
mkdir -p gpurun_out/r3_run10
O=gpurun_out/r3_run10
FNV_FULLSIZE=0 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_device_build.py tests/test_gpu_python_api.py tests/test_gpu_multi_device.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python tools/latency_probe.py > $O/latency.txt 2>&1; grep -v amdgpu $O/latency.txt
python tools/latency_probe.py 1000000 shadow_exact=0 > $O/latency_noshadow.txt 2>&1; grep -v amdgpu $O/latency_noshadow.txt | grep -E "search_single|batch +(64|256|1024)"
