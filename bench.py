#!/usr/bin/env python3
"""bench.py -- QPS of the batched flat-NSW k-NN search on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch: `nq` queries (default 10 000) searched
against an index resident in HBM (default: the SIFT-1M stand-in of SURVEY.md 8d -- 1M x 128
float32, integer-valued 0..255, L2, M=32, ef_construction=100, ef_search=100, K=10; SIFT itself
cannot be downloaded here).  Queries and result buffers live in HBM when the timed region starts.

  python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel
= beam_search_kernel; achieved = algorithmic bytes per launch / average launch duration from HIP
events on the launch stream) and, at N=1, `cpu_baseline` (the CPU oracle timed on the host cores).
Multi-GPU: index replicated with one RCCL broadcast per buffer at load, queries sharded, no
per-query collective; weak scaling (every rank searches its own nq queries).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def recorded_traffic(n, nq, ef):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes); None unless this run is the
    very workload those passes profiled."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    for r in rec if isinstance(rec, list) else [rec]:
        if (r.get("n"), r.get("nq"), r.get("ef")) == (n, nq, ef):
            return r.get("hbm_bytes_per_launch_corrected")
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--index-size", dest="n", type=int, default=1_000_000, help="number of indexed vectors")
    ap.add_argument("--nq", type=int, default=10_000, help="queries per batch per GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--efc", type=int, default=100)
    ap.add_argument("--ef", type=int, default=0,
                    help="ef_search; 0 = the metric's rule: smallest ef of --ef-sweep with recall@10 >= 0.95")
    ap.add_argument("--ef-sweep", default="50,52,54,56,58,60,64,70,80,100,150,200,400")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--dtype", default="float32", choices=["float32", "uint8"],
                    help="index element type (uint8: the same integer-valued data stored as bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="C-ABI option name=value")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    # BENCH_SHARE_GPU=1 (testing only): all ranks use cuda:0 and talk over gloo, so the multi-rank control flow
    # can be exercised on a single-GPU box; the real multi-GPU run is one rank per GPU over RCCL.
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import flatnav_amd as flatnav
    from flatnav_amd import datasets as ds
    from flatnav_amd import hip

    N, NQ, DIM, M, K = args.n, args.nq, args.dim, args.M, args.K
    hw = ds.effective_cpus()  # honours the cgroup CPU quota (16 on the MI355X boxes, 256 CPUs visible)

    # ---- data: every rank generates the same base stream; queries differ per rank (sharding) ----
    t0 = time.time()
    X, Q_all = ds.sift_like(N, NQ * world, dim=DIM)
    Q = np.ascontiguousarray(Q_all[rank * NQ:(rank + 1) * NQ])
    DT = args.dtype
    ESIZE = 4 if DT == "float32" else 1
    if DT == "uint8":
        X, Q = X.astype(np.uint8), Q.astype(np.uint8)
    log("[rank %d] data %.1fs" % (rank, time.time() - t0))

    # ---- index: rank 0 builds with the product's host builder, uploads, broadcasts ---------------
    blob_info = None
    index = None
    if rank == 0:
        t0 = time.time()
        index = flatnav.index.create(distance_type="l2", index_data_type=getattr(flatnav.data_type.DataType, DT),
                                     dim=DIM, dataset_size=N, max_edges_per_node=M)
        threads = args.build_threads or max(1, min(hw + hw // 2, os.cpu_count() or 1))
        index.set_num_threads(threads)
        index.add(data=X, ef_construction=args.efc)
        log("[rank 0] host build: %d nodes, %d threads, %.1fs" % (N, threads, time.time() - t0))
        t0 = time.time()
        blob = np.asarray(index._raw_blob())
        dev = hip.DeviceIndex.upload(blob, index._node_size_bytes, index._data_size_bytes, M, N, DT, "l2", DIM,
                                     device=local_rank)
        log("[rank 0] upload + re-layout to HBM: %.2fs" % (time.time() - t0))
    else:
        dev = hip.DeviceIndex.alloc(M, N, DT, "l2", DIM, device=local_rank)
    if world > 1:
        from flatnav_amd import multigpu

        t0 = time.time()
        multigpu.replicate_index(dev, local_rank, src=0)  # one RCCL broadcast per buffer over xGMI, at load only
        log("[rank %d] index broadcast %.2fs" % (rank, time.time() - t0))
    for o in args.opt:
        k, v = o.split("=")
        dev.set_option(k, int(v))

    # ---- device-resident inputs / outputs ---------------------------------------------------------
    dq = torch.from_numpy(Q).cuda()
    d_dist = torch.empty((NQ, K), dtype=torch.float32, device="cuda")
    d_lab = torch.empty((NQ, K), dtype=torch.int32, device="cuda")
    d_cnt = torch.empty(NQ, dtype=torch.int32, device="cuda")
    d_nd = torch.zeros(NQ, dtype=torch.int64, device="cuda")
    d_nh = torch.zeros(NQ, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()

    # ---- exact ground truth for recall@10 (brute force on the GPU, first 1000 queries of this rank) ----
    nrec = min(1000, NQ)
    xt = torch.from_numpy(X).cuda().float()
    qt = dq[:nrec].float()
    xn = (xt * xt).sum(1)
    gt = torch.empty((nrec, K), dtype=torch.int64, device="cuda")
    for s0 in range(0, nrec, 250):
        d2 = xn[None, :] - 2.0 * (qt[s0:s0 + 250] @ xt.T)
        gt[s0:s0 + 250] = torch.topk(d2, K, dim=1, largest=False).indices
    gt = gt.cpu().numpy()
    del xt, xn, qt
    torch.cuda.empty_cache()

    def recall_at(ef):
        dev.search_device(dq.data_ptr(), nrec, K, ef, 100, d_dist.data_ptr(), d_lab.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        dev.status()
        return ds.recall_at_k(d_lab[:nrec].cpu().numpy(), gt)

    # ---- ef_search: the metric is QPS at recall@10 >= 0.95 -> smallest swept ef that reaches it (rank 0 decides) ----
    sweep = {}
    if args.ef > 0:
        EF = args.ef
    else:
        EF = 0
        for ef in sorted(int(x) for x in args.ef_sweep.split(",")):
            if rank == 0:
                sweep[ef] = round(recall_at(ef), 4)
                ok = sweep[ef] >= 0.95
            else:
                ok = False
            if dist is not None:
                flag = torch.tensor([1 if ok else 0], device="cuda")
                dist.broadcast(flag, src=0)
                ok = bool(flag.item())
            if ok:
                EF = ef
                break
        if EF == 0:
            EF = max(int(x) for x in args.ef_sweep.split(","))
        log("[rank %d] ef sweep %s -> ef_search=%d" % (rank, sweep, EF))

    def step():
        dev.search_device(dq.data_ptr(), NQ, K, EF, 100, d_dist.data_ptr(), d_lab.data_ptr(), d_cnt.data_ptr(),
                          d_nd.data_ptr(), d_nh.data_ptr(), stream=stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dev.status()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev.status()
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic bytes of one launch from the run's own counters --------------
    nd = d_nd.cpu().numpy().astype(np.int64)
    nh = d_nh.cpu().numpy().astype(np.int64)
    step_nodes = max(1, N // 100)
    n_scan = (N + step_nodes - 1) // step_nodes
    bytes_launch = int(((n_scan + nd) * DIM * ESIZE + nh * M * 4 + K * 4).sum())
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = bytes_launch / avg_kernel_s / 1e9

    out = None
    if rank == 0:
        labels = d_lab.cpu().numpy()
        recall = ds.recall_at_k(labels[:nrec], gt)
        geom = dev.launch_geometry()
        # informational: the host-buffer entry point (pageable H2D of the queries + kernel + D2H of results)
        t0 = time.perf_counter()
        for _ in range(3):
            dev.search(Q, K, EF)
        host_qps = 3 * NQ / (time.perf_counter() - t0)
        log("[rank 0] host-buffer (PCIe-inclusive) path: %.0f queries/s" % host_qps)
        total_q = NQ * world * args.steps
        out = {
            "metric": "qps_at_recall10_ge_0.95",
            "value": total_q / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if DT == "float32" else "u8",
            "data": "synthetic",
            "config": {
                "workload": "SIFT-1M stand-in (S1 int-lowrank, SURVEY.md 8d): %d x %d %s L2, M=%d, "
                            "ef_construction=%d, ef_search=%d, K=%d, %d batched queries per GPU, index in HBM"
                            % (N, DIM, DT, M, args.efc, EF, K, NQ),
                "recall_at_10": round(recall, 4),
                "ef_search": EF,
                "ef_selection": ("fixed by --ef" if args.ef > 0 else
                                 "smallest ef of the sweep with recall@10 >= 0.95 (SURVEY.md 8d); recalls: %s" % sweep),
                "parallelism": "index replicated x%d, queries sharded" % world,
                "mean_dist_evals_per_query": float(nd.mean()),
                "mean_hops_per_query": float(nh.mean()),
                "launch": geom,
                "host_buffer_qps_pcie_inclusive": round(host_qps),
            },
            "roofline": {
                "bound": "hbm",
                # name as rocprofv3 prints it: <element type, metric 0=L2, G lanes per vector, CU loads, FULL rows>
                "kernel": "fnv_dev::beam_search_kernel<%s, 0, 8, %d, true>" % ("float" if DT == "float32" else "unsigned char",
                                                                     4 if DT == "float32" else 1),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": recorded_traffic(N, NQ, EF),
                "algorithmic_bytes_per_launch": bytes_launch,
                "avg_kernel_ms": avg_kernel_s * 1e3,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(index, Q, K, EF, hw, labels, DT)
        del index
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline(index, Q, K, EF, hw, gpu_labels, dtype="float32"):
    """The CPU oracle (restated reference search, oracle/) on the same graph and queries, all host
    threads, bounded to roughly 10-20 s.  Also re-checks GPU == CPU ids on the sample."""
    from oracle import oracle as orc

    orc.build()
    blob = np.asarray(index._raw_blob())
    n = int(index._cur_num_nodes)
    o = orc.OracleIndex.from_blob("l2", dtype, Q.shape[1], n, n, index.max_edges_per_node, blob)
    kind_note = "oracle port, own AVX2 distance"
    if o.use_reference_distance(True):
        kind_note = "oracle port of Index::search driving the reference's own compiled AVX-512 distance kernel (oracle/_ref)"
    threads = hw
    o.search(Q[:256], K, EF, threads=threads)  # warm
    reps, t_used, nq_done = 0, 0.0, 0
    ol = None
    while t_used < 8.0 and reps < 50:
        t0 = time.perf_counter()
        _, ol = o.search(Q, K, EF, threads=threads)
        t_used += time.perf_counter() - t0
        nq_done += len(Q)
        reps += 1
    t0 = time.perf_counter()
    o.search(Q[:2000], K, EF, threads=1)
    qps1 = 2000 / (time.perf_counter() - t0)
    same = float((ol == gpu_labels).all(axis=1).mean())
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": nq_done / t_used,
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d x the same %d-query batch on %d host threads (%s); single-thread: %.0f queries/s; "
                  "GPU ids == CPU ids on %.2f%% of queries; host: %s, %d CPUs visible, %d usable (cgroup quota)"
                  % (reps, len(Q), threads, kind_note, qps1, same * 100, cpu_model, os.cpu_count() or 0, hw),
    }


if __name__ == "__main__":
    main()
