#!/usr/bin/env python3
"""Developer tool: two 10 000-query launches in flight (two handles on the same device buffers, two streams, launches
alternate) against one launch at a time -- for the in-tree library and, optionally, an older build (--lib name=path) on the
same index.  The rate with two launches in flight is what a caller who always has the next batch ready gets,
and the measure of what a single launch loses to its ramp and drain.

  python tools/dev/pipelined_probe.py --config c2 [--dtype uint8] [--ef 52] [--lib r4=flatnav_amd/_exp/libflatnav_hip_r4.so]
"""
import argparse, ctypes, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa: E402
import bench, flatnav_amd as flatnav  # noqa: E402
from flatnav_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--ef", type=int, default=52)
ap.add_argument("--lib", default="")
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
cfg = dict(bench.CONFIGS[args.config]); N = cfg["n"]; DIM = cfg["dim"]; DT = args.dtype; NQ, NB, K, M = 10000, 8, 10, 32
dev_t = torch.device("cuda", 0); torch.cuda.set_device(0)
data = bench.Data(cfg, N, NQ * NB, torch, dev_t)
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=getattr(flatnav.data_type.DataType, DT), dim=DIM, dataset_size=N, max_edges_per_node=M)
index.set_num_threads(16); index.set_device(0)
for first, xh in data.chunks(5_000_000):
    index.add(data=xh.astype(np.uint8) if DT == "uint8" else xh, ef_construction=100, labels=list(range(first, first + len(xh))), device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
Q = data.queries(); Q = Q.astype(np.uint8) if DT == "uint8" else Q
dq = torch.from_numpy(np.ascontiguousarray(Q).reshape(NB, NQ, DIM)).to(dev_t)
mods = {"tree": (hip, dev)}
if args.lib:
    name, path = args.lib.split("=")
    os.environ["FLATNAV_HIP_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("flatnav_amd.hip_" + name, os.path.join(ROOT, "flatnav_amd", "hip.py"))
    m = importlib.util.module_from_spec(spec); sys.modules[spec.name] = m; spec.loader.exec_module(m)
    del os.environ["FLATNAV_HIP_LIB"]
    mods[name] = (m, m.DeviceIndex.adopt(dev.device_buffers(), M, N, DT, cfg["metric"], DIM, device=0, keep_alive=index))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
outs = [(torch.empty((NQ, K), dtype=torch.float32, device=dev_t), torch.empty((NQ, K), dtype=torch.int32, device=dev_t)) for _ in range(2)]
for rnd in range(2):
    for name, (m, d) in mods.items():
        v = d.view()
        for h in (d, v):
            h.tune(int(dq[0].data_ptr()), K, args.ef, 100, nq=NQ)
        def run(n, lanes):
            for i in range(n):
                h, st, (od, ol) = lanes[i % len(lanes)]
                h.search_device(dq[i % NB].data_ptr(), NQ, K, args.ef, 100, od.data_ptr(), ol.data_ptr(), stream=st.cuda_stream)
        res = {}
        for label, lanes in (("one at a time", [(d, s1, outs[0])]), ("two in flight", [(d, s1, outs[0]), (v, s2, outs[1])])):
            run(4, lanes); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(args.steps, lanes); torch.cuda.synchronize()
            res[label] = NQ * args.steps / (time.perf_counter() - t0)
        print("%-5s round %d: one at a time %.3f M queries/s, two in flight %.3f M (x %.2f); variants %s / %s" % (
            name, rnd, res["one at a time"] / 1e6, res["two in flight"] / 1e6, res["two in flight"] / res["one at a time"],
            d.launch_info()["variant"], v.launch_info()["variant"]), flush=True)
        v.status(); v.close()
