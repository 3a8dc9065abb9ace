#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run45; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -k "ground_truth" > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-300
