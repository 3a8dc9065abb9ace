#!/bin/bash
# round 3, run 21 (final library): the whole GPU suite on the final library, then the round's profile set (bench line, rocprofv3 traces, PMC passes)
mkdir -p gpurun_out/r3_run21
( time python -m pytest tests -m gpu -q ) > gpurun_out/r3_run21/pytest.log 2>&1
tail -6 gpurun_out/r3_run21/pytest.log
bash tools/collect_profiles.sh > gpurun_out/r3_run21/collect.log 2>&1
tail -5 gpurun_out/r3_run21/collect.log
# keep what summarise_profiles.py reads, drop the bulky raw traces
cd gpurun_out/profile_set && find . -name "*_kernel_trace.csv" -size +20M -delete; du -sh .
