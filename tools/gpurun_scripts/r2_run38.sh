#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run38; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "random_shapes" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
