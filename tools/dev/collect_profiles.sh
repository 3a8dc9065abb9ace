#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/profile_set (run through gpurun from the repo root):
#   1. the default bench invocation (the contract line + every further configuration);
#   2. per configuration, at the ef that run selected: rocprofv3 --kernel-trace --stats of a bench command (the timed
#      region's launches), and separate --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE);
#   3. SQ instruction-mix / wait counters for the float32 headline and the uint8 index.
# tools/dev/summarise_profiles.py <tag> then condenses them into profiles/.
#   usage: collect_profiles.sh [configs...]     (default: all seven)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profile_set
mkdir -p $O
CONFIGS=${@:-c2 c2-uint8 c4 c3-lowrank c3 c5 c5-lowrank}
if [ ! -s $O/bench.json ]; then
  python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log
fi
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
  EF=$(python3 -c "
import json
d = json.load(open('$O/bench.json'))
e = d if '$C' == 'c2' else d['$C']
print(e['config']['ef_search'])")
  STEPS=20; case $C in c3*|c5*) STEPS=6;; esac
  ARGS="--config $C --ef $EF --no-cpu-baseline --no-secondary --sustain-seconds 0 --warmup 3"
  echo "== $C ef=$EF"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$C -o bench -- python3 $R/bench.py $ARGS --steps $STEPS > $O/trace_$C.json 2> $O/trace_$C.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 > $O/fetch_$C.json 2> /dev/null
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 > $O/write_$C.json 2> /dev/null
  case $C in c2|c2-uint8)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 > $O/sq_$C.json 2> /dev/null;;
  esac
done
ls $O | head -60
