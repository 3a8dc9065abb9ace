#!/bin/bash
# round 4, run 12: the round's profile set from the FINAL library (default bench invocation; per configuration a kernel
# trace of the timed region and FETCH_SIZE / WRITE_SIZE passes; SQ counters for the float32 headline and the uint8 index);
# raw rocprofv3 files reduced to the search kernels' rows on the box; then the whole GPU suite on the same library
rm -rf gpurun_out/profile_set
bash tools/dev/collect_profiles.sh > gpurun_out/profile_set.log 2>&1
tail -12 gpurun_out/profile_set.log
cat gpurun_out/profile_set/summary.log | tail -20
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/final_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/final_pytest.log
tail -45 gpurun_out/final_pytest.log
du -sh gpurun_out
