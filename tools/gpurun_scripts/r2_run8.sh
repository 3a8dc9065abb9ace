#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run8; mkdir -p $O
timeout 1200 python tools/latency_probe.py > $O/latency.txt 2>&1; cat $O/latency.txt | tail -30
timeout 2400 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_bench.py tests/test_cpp_api.py tests/test_gpu_python_api.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 2400 python tools/builder_compare.py 2000000 > $O/builder_compare.txt 2>&1; tail -5 $O/builder_compare.txt
