#!/bin/bash
# round 3, run 3: instruction diet of the merged-beam hop (visited probe, vote masks, single-chunk path): parity + effect
mkdir -p gpurun_out/r3_run3
O=gpurun_out/r3_run3
FNV_FULLSIZE=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_device_build.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
QUICK="--no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 20 --warmup 5"
for a in "--dtype float32" "--dtype uint8" "--dtype float32 --ef 100" "--dtype uint8 --ef 100" "--config c4 --ef 110" "--config c4 --ef 200" "--config c4 --ef 400"; do
  python bench.py $QUICK $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],3), d['config']['launch'], d['config']['kernel_variant'], d['config']['queries_replayed_by_exact_kernel'])" >> $O/bench_lines.txt 2>&1
done
cat $O/bench_lines.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dt in float32 uint8; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/$O/sq_$dt -o bench -- python3 $R/bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 --ef 52 --dtype $dt --steps 3 --warmup 5 > /dev/null 2>&1
done
cd $R
python - <<'P'
import csv,glob,collections
for dt in ("float32","uint8"):
    f=glob.glob("gpurun_out/r3_run3/sq_%s/**/*counter_collection.csv"%dt, recursive=True)
    if not f: print(dt,"no counters"); continue
    rows=[r for r in csv.DictReader(open(f[0])) if "merged" in r["Kernel_Name"]]
    g=max(int(r["Grid_Size"]) for r in rows)
    rows=[r for r in rows if int(r["Grid_Size"])==g]
    ids=sorted(set(int(r["Dispatch_Id"]) for r in rows))[-3:]
    acc=collections.Counter()
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(dt, {k:round(v/3/1e6,1) for k,v in acc.items()})
P
