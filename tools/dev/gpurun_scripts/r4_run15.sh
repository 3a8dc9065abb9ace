#!/bin/bash
# round 4, run 15: the lane pool -- tests, then single queries from 1 ... 16 caller threads
O=gpurun_out/r4_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_python_api.py tests/test_gpu_multi_device.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/dev/latency_probe.py > $O/latency.txt 2>$O/latency.err; cat $O/latency.txt
