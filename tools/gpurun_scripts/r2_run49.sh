#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run49; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi_device.py -x -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-300
