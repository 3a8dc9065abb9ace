#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run9; mkdir -p $O
timeout 1200 python tools/latency_probe.py > $O/latency.txt 2>&1; cat $O/latency.txt | tail -30
timeout 900 python tools/r2_probe.py time 200000 s3 > $O/time_s3.txt 2>&1; tail -30 $O/time_s3.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
