#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/profile_set (run through gpurun from the repo root):
#   1. the default bench invocation (the contract line + every further configuration);
#   2. per configuration, at the ef that run selected: rocprofv3 --kernel-trace --stats of a bench command (the timed
#      region's launches), and separate --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE);
#   3. SQ instruction-mix / wait counters for the float32 headline and the two uint8 indexes.
# tools/dev/summarise_profiles.py <tag> then condenses them into profiles/.
#   usage: collect_profiles.sh [configs...]     (default: all eight)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profile_set
mkdir -p $O
CONFIGS=${@:-c2 c2-uint8 c4 c3-lowrank c3 c5 c5-lowrank c5-uint8}
# (a GPU box starts without gpurun_out/: later calls for a subset of the configurations take the efs from the bench line
#  that an earlier call left -- copied to profiles/<tag>_bench.json, which travels with the repo)
if [ ! -s $O/bench.json ] && [ -s $R/profiles/${PROFILE_TAG:-r6}_bench.json ] && [ -n "${PROFILE_REUSE_BENCH:-}" ]; then
  cp $R/profiles/${PROFILE_TAG:-r6}_bench.json $O/bench.json
fi
if [ ! -s $O/bench.json ]; then
  # (round 5: stdout is the <= 4 KB contract line; the full record -- what the steps below read -- goes to --full-record)
  python $R/bench.py --steps 20 --warmup 5 --full-record $O/bench.json > $O/bench_line.json 2> $O/bench.log
fi
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
  EF=$(python3 -c "
import json
d = json.load(open('$O/bench.json'))
e = d if '$C' == 'c2' else d['$C']
print(e['config']['ef_search'])")
  STEPS=20; case $C in c3*|c5*) STEPS=6;; esac
  ARGS="--config $C --ef $EF --no-cpu-baseline --no-secondary --sustain-seconds 0 --warmup 3 --regions 1"
  echo "== $C ef=$EF"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$C -o bench -- python3 $R/bench.py $ARGS --steps $STEPS --full-record $O/trace_$C.json > /dev/null 2> $O/trace_$C.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 --full-record $O/fetch_$C.json > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 --full-record $O/write_$C.json > /dev/null 2>&1
  case $C in c2|c2-uint8|c5-uint8)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq_$C -o bench -- python3 $R/bench.py $ARGS --steps 3 --full-record $O/sq_$C.json > /dev/null 2>&1;;
  esac
  # gpurun brings back 64 MB and the raw rocprofv3 output of a 50M-node build is hundreds: keep the search kernels' rows of
  # the traces / counter files and the --stats table, drop the rest; summarise here as well (summary/<config>.json)
  for D in $O/trace_$C $O/fetch_$C $O/write_$C $O/sq_$C; do
    [ -d $D ] || continue
    find $D -type f \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' \) | while read F; do
      { head -1 "$F"; grep beam_search "$F"; } > "$F.small" && mv "$F.small" "$F"
    done
    find $D -type f ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' -delete
  done
  python3 $R/tools/dev/summarise_profiles.py --one $C >> $O/summary.log 2>&1
  tail -2 $O/summary.log
done
du -sh $O; ls $O $O/summary | head -60
