#!/bin/bash
# round 3, run 14: deep randomised parity sweeps on the final kernels (search forms vs exact kernel vs oracle; builders; API sequences)
# + fresh SQ counters of the timed region (float32 / uint8)
mkdir -p gpurun_out/r3_run14
O=gpurun_out/r3_run14
FNV_FULLSIZE=0 FNV_FUZZ_TRIALS=400 FNV_FUZZ_SEED=31 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -q -k "random" > $O/fuzz.log 2>&1
echo "fuzz rc=$?" >> $O/fuzz.log
tail -5 $O/fuzz.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dt in float32 uint8; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/$O/sq_$dt -o bench -- python3 $R/bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 --ef 52 --dtype $dt --steps 3 --warmup 5 > /dev/null 2>&1
done
cd $R
python - <<'P'
import csv,glob,collections,json
out={}
for dt in ("float32","uint8"):
    f=glob.glob("gpurun_out/r3_run14/sq_%s/**/*counter_collection.csv"%dt, recursive=True)
    rows=[r for r in csv.DictReader(open(f[0])) if "beam_search" in r["Kernel_Name"]]
    ids=sorted(set(int(r["Dispatch_Id"]) for r in rows))[-3:]
    rows=[r for r in rows if int(r["Dispatch_Id"]) in ids]
    acc=collections.defaultdict(list)
    for r in rows: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    first=rows[0]
    out[dt]={"kernel":{k:first[k] for k in ("Grid_Size","Workgroup_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Kernel_Name")},
             "per_launch":{k:sum(v)/len(v) for k,v in acc.items()}}
    print(dt,{k:round(sum(v)/len(v)/1e6,1) for k,v in acc.items()})
json.dump({"ef":52,"counters":out},open("gpurun_out/r3_run14/r3_sq_counters.json","w"),indent=1)
P
rm -rf $O/sq_float32 $O/sq_uint8
