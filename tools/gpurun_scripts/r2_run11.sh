#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run11; mkdir -p $O
timeout 900 python tools/r2_probe.py reasons sift > $O/reasons_sift.txt 2>&1; cat $O/reasons_sift.txt
timeout 900 python tools/r2_probe.py reasons glove > $O/reasons_glove.txt 2>&1; cat $O/reasons_glove.txt
