#!/usr/bin/env python3
"""Developer tool: static instruction counts between the ISA markers of the bench kernel
(build with -DFNV_ASM_MARKS).  Straight-line counts only -- loops are listed with their body size."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/isa_regions.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DFNV_ASM_MARKS",
                       "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                       os.path.join(ROOT, "flatnav_amd/csrc/beam_search.hip"), "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
name = "_ZN7fnv_dev18beam_search_kernelIfLi0ELi8ELi4ELb1EEEvNS_12SearchParamsE"
i = s.find(name + ":"); j = s.find(".Lfunc_end", i)
lines = s[i:j].splitlines()
cur = "start"; counts = collections.OrderedDict(); kinds = collections.defaultdict(collections.Counter)
for l in lines:
    t = l.strip()
    m = re.match(r";\s*##MARK (\S+)", t)
    if m:
        cur = m.group(1); continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    counts[cur] = counts.get(cur, 0) + 1
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    kinds[cur][k] += 1
for k, v in counts.items():
    print("%-12s %5d  %s" % (k, v, dict(kinds[k])))
