#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run15; mkdir -p $O
timeout 1200 python tools/latency_probe.py > $O/latency.txt 2>&1; grep -v amdgpu $O/latency.txt | grep -v "batch  *[0-9]*[46]:" | tail -40
