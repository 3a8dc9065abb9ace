#!/bin/bash
# round 3, run 22: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the 30 GB configurations: c3-lowrank (10M x 768, ef=800)
# and c5-lowrank (50M x 128, ef=80) -- tables the 256 MiB Infinity Cache cannot help
mkdir -p gpurun_out/r3_run22
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_run22
cd /tmp && export TMPDIR=/tmp
for cfg in "c3-lowrank 800" "c5-lowrank 80"; do
  set -- $cfg
  ARGS="--config $1 --no-cpu-baseline --no-secondary --secondary-configs none --sustain-seconds 0 --ef $2 --steps 3 --warmup 2"
  python3 $R/bench.py $ARGS > $O/bench_$1.json 2> $O/bench_$1.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$1 -o bench -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$1 -o bench -- python3 $R/bench.py $ARGS > /dev/null 2>&1
done
cd $R
python - <<'P'
import csv,glob,json,collections
out=[]
for name,ef in (("c3-lowrank",800),("c5-lowrank",80)):
    b=json.load(open("gpurun_out/r3_run22/bench_%s.json"%name))
    vals={}
    for kind in ("fetch","write"):
        f=glob.glob("gpurun_out/r3_run22/%s_%s/**/*counter_collection.csv"%(kind,name), recursive=True)[0]
        rows=[r for r in csv.DictReader(open(f)) if "beam_search" in r["Kernel_Name"]]
        ids=sorted({int(r["Dispatch_Id"]) for r in rows})[-3:]
        rows=[r for r in rows if int(r["Dispatch_Id"]) in ids]
        vals[kind]=sum(float(r["Counter_Value"]) for r in rows)/3
        kern=rows[-1]["Kernel_Name"]
    r=b["roofline"]
    corrected=(2*vals["fetch"]+vals["write"])*1024
    out.append({"config":name,"dtype":"float32","n":None,"nq":10000,"ef":ef,"kernel":kern,
                "FETCH_SIZE_KB_per_launch":vals["fetch"],"WRITE_SIZE_KB_per_launch":vals["write"],
                "hbm_bytes_per_launch_corrected":corrected,"algorithmic_bytes_per_launch":r["algorithmic_bytes_per_launch"],
                "traffic_over_algorithmic":corrected/r["algorithmic_bytes_per_launch"],
                "avg_kernel_ms":r["avg_kernel_ms"],"frac":r["frac"],"value":b["value"]})
    print(name, "traffic %.1f GB, algorithmic %.1f GB, ratio %.3f, kernel %.2f ms, frac %.3f" % (corrected/1e9, r["algorithmic_bytes_per_launch"]/1e9, corrected/r["algorithmic_bytes_per_launch"], r["avg_kernel_ms"], r["frac"]))
json.dump(out,open("gpurun_out/r3_run22/pmc_big_tables.json","w"),indent=1)
P
rm -rf $O/fetch_* $O/write_*
