import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sys, re, subprocess
# usage: resusage.py <defines...>  -> table of kernel resource usage for a fast dev build
from flatnav_amd import build as hb
import io, contextlib, os
defs = sys.argv[1:]
cmd = [hb.hipcc(), "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-inline-asm",
       "-fvisibility=hidden", "-I" + os.path.join(hb.ROOT, "include")] + ["-D" + d for d in defs] + ["-c", hb.MAIN, "-o", "/tmp/resusage.o"]
p = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: 
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":",1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":",1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    if "fnv_dev" not in k or "relayout" in k or "gather_ceiling" in k or "iota" in k or "scatter_links" in k:
        continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(fnv_dev::SearchParams\)|void fnv_dev::", "", name)
    print("%-70s VGPR %s AGPR %s SGPR %s spillS %s spillV %s scratch %s occ %s LDS %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
