#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run24; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
