#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/profile_set (run through gpurun from the repo root):
#   the bench line, rocprofv3 --kernel-trace --stats of the same command, separate --pmc passes for HBM traffic
#   (FETCH_SIZE / WRITE_SIZE) and SQ instruction-mix / wait counters, for the float32 headline and the uint8 index.
# tools/summarise_profiles.py <tag> then condenses them into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profile_set
rm -rf $O; mkdir -p $O
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log
EF=$(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['ef_search'])")
echo "selected ef=$EF"
cd /tmp && export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-secondary --secondary-configs none --sustain-seconds 0 --ef $EF"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py $QUICK --steps 20 --warmup 5 > $O/trace.log 2>&1
for dt in float32 uint8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$dt -o bench -- python3 $R/bench.py $QUICK --dtype $dt --steps 3 --warmup 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$dt -o bench -- python3 $R/bench.py $QUICK --dtype $dt --steps 3 --warmup 5 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_$dt -o bench -- python3 $R/bench.py $QUICK --dtype $dt --steps 3 --warmup 5 > /dev/null 2>&1
done
# 100-d rows (config c4) at its recall-rule ef and at wide beams: traffic against algorithmic and against line bytes
for EFW in 110 200 400; do
  WIDE="--config c4 --no-cpu-baseline --no-secondary --sustain-seconds 0 --ef $EFW"
  python3 $R/bench.py $WIDE --steps 5 --warmup 3 > $O/bench_c4_ef$EFW.json 2> $O/bench_c4_ef$EFW.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c4_ef$EFW -o bench -- python3 $R/bench.py $WIDE --steps 3 --warmup 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c4_ef$EFW -o bench -- python3 $R/bench.py $WIDE --steps 3 --warmup 3 > /dev/null 2>&1
done
python $R/bench.py --dtype uint8 --steps 20 --warmup 5 > $O/bench_uint8.json 2> $O/bench_uint8.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_uint8 -o bench -- python3 $R/bench.py $QUICK --dtype uint8 --steps 20 --warmup 5 > $O/trace_uint8.log 2>&1
ls $O $O/trace | head -40
