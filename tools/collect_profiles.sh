#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/ (run through gpurun from the repo root):
#   bench line, rocprofv3 --kernel-trace --stats of the same command, separate --pmc passes for HBM traffic
#   (FETCH_SIZE / WRITE_SIZE, per swept ef) and SQ instruction-mix counters.
# tools/summarise_profiles.py then condenses them into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profile_set
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.log
EF=$(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['ef_search'])")
echo "selected ef=$EF"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 5 --ef $EF > $O/trace.log 2>&1
for ef in $(echo 50 60 $EF | tr ' ' '\n' | sort -un); do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_ef$ef -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --ef $ef > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_ef$ef -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --ef $ef > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --ef $EF > /dev/null 2>&1
ls $O
