#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run56; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/latency_probe.py 1000000 2>&1 | grep -v amdgpu > $O/latency.txt; grep search_single $O/latency.txt
