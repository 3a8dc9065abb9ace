#!/usr/bin/env python3
"""Developer tool: static instruction counts between the phase markers of one search-kernel instantiation
(-DFNV_ASM_MARKS: every PH_MARK(i) of the kernels drops a named comment into the ISA).

  python tools/dev/isa_regions.py [--family 5] [--type float] [--tag f32] [--metric 0] [--kernel Li8ELi4ELb1ELi1E] [--defines ...]

family: kernel_inst.hip's FNV_INST_FAMILY (0 exact two-heap kernel, 4 / 7 / 5 merged beam in 4 / 2 / 1 register chunks, 6 merged
beam in LDS); --kernel: a substring of the mangled name that picks the instantiation (default: G=8, CU=4, FULL, R=1 -- the
1M x 128 float32 bench kernel).  Markers (merged_beam.hpp): phase0 query staged, 1 entry point found, 2/3 node selected (hop
starts), 4 visited test done, 5 distances done, 6 merge done, 7 query finished.  A region is everything between a marker
and the next one IN PROGRAM ORDER, so the first regions after a loop head also hold the code the compiler moved there; loops
inside a region are listed with their body sizes (a static count: the dynamic count is body x trips).
"""
import argparse
import collections
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--family", type=int, default=5)
ap.add_argument("--type", default="float")
ap.add_argument("--tag", default="f32")
ap.add_argument("--metric", type=int, default=0)
ap.add_argument("--kernel", default="Li8ELi4ELb1ELi1E")
ap.add_argument("--defines", nargs="*", default=[])
args = ap.parse_args()

out = "/tmp/isa_regions.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DFNV_ASM_MARKS", "-Wno-unused-function", "-Wno-inline-asm",
       "-I" + os.path.join(ROOT, "include"), "-DFNV_INST_T=" + args.type, "-DFNV_INST_TAG=" + args.tag, "-DFNV_INST_METRIC=%d" % args.metric,
       "-DFNV_INST_MTAG=" + ("l2" if args.metric == 0 else "ip"), "-DFNV_INST_FAMILY=%d" % args.family] + ["-D" + d for d in args.defines] + [
           "-S", "--cuda-device-only", os.path.join(ROOT, "flatnav_amd/csrc/kernel_inst.hip"), "-o", out]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
s = open(out).read()
names = [m.group(1) for m in re.finditer(r"^(_ZN7fnv_dev\S*%s\S*):" % re.escape(args.kernel), s, re.M)]
if not names:
    raise SystemExit("no kernel matches %r; candidates:\n  %s" % (args.kernel, "\n  ".join(sorted(set(re.findall(r"^(_ZN7fnv_dev\S+):", s, re.M))))))
name = names[0]
i = s.find(name + ":")
j = s.find(".Lfunc_end", i)
lines = s[i:j].splitlines()
print("# %s  (%d ISA lines)" % (subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip(), len(lines)))


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


cur, order = "start", []
regions = collections.OrderedDict()
for l in lines:
    t = l.strip()
    m = re.match(r";\s*##MARK (\S+)", t)
    if m:
        cur = "%s#%d" % (m.group(1), sum(1 for r in regions if r.startswith(m.group(1) + "#")))
        continue
    if not t or t.startswith((";", ".", "//")):
        continue
    if t.endswith(":"):
        continue
    op = t.split()[0]
    regions.setdefault(cur, collections.Counter())[kind(op)] += 1
print("%-12s %6s   %s" % ("after marker", "instr", "valu / salu / branch / wait / lds / vmem"))
for r, c in regions.items():
    n = sum(c.values())
    print("%-12s %6d   %4d / %4d / %4d / %4d / %4d / %4d" % (r, n, c["valu"], c["salu"], c["branch"], c["wait"], c["lds"], c["vmem"]))
