#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run10; mkdir -p $O
timeout 900 python tools/phase_profile.py --ef 52 > $O/phase_ef52.txt 2>&1; cat $O/phase_ef52.txt
timeout 900 python tools/phase_profile.py --ef 100 > $O/phase_ef100.txt 2>&1; cat $O/phase_ef100.txt | tail -80
