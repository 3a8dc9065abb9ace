#!/bin/bash
# round 3, run 42: whole GPU suite again (reference-order bar), then the uint8 part of the profile set with the guess off for 1-byte rows
mkdir -p gpurun_out/r3_run42
O=gpurun_out/r3_run42
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
R=$PWD
P=$R/gpurun_out/profile_set
mkdir -p $P
python $R/bench.py --dtype uint8 --steps 20 --warmup 5 > $P/bench_uint8.json 2> $P/bench_uint8.log
EF=$(python3 -c "import json;print(json.load(open('$P/bench_uint8.json'))['config']['ef_search'])")
cd /tmp && export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-secondary --secondary-configs none --sustain-seconds 0 --ef $EF"
rm -rf $P/sq_uint8 $P/trace_uint8 $P/fetch_uint8 $P/write_uint8
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch_uint8 -o bench -- python3 $R/bench.py $QUICK --dtype uint8 --steps 3 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write_uint8 -o bench -- python3 $R/bench.py $QUICK --dtype uint8 --steps 3 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/sq_uint8 -o bench -- python3 $R/bench.py $QUICK --dtype uint8 --steps 3 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_uint8 -o bench -- python3 $R/bench.py $QUICK --dtype uint8 --steps 20 --warmup 5 > $P/trace_uint8.log 2>&1
cd $R
python3 -c "
import json; d=json.load(open('$P/bench_uint8.json')); print('uint8', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['kernel_variant'], (d.get('pipelined') or {}).get('value'))"
