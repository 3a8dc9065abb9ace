#!/usr/bin/env python3
"""bench.py -- QPS of the batched flat-NSW k-NN search on MI355X, one line per BASELINE.json configuration.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2-uint8|c3|c3-lowrank|c4|c5|c5-lowrank|c5-uint8]
                  [--secondary-configs c4,c3-lowrank,c5,c5-lowrank | none] [--index-size n ...]

One "step" = one pass of the hot path over one batch: `nq` queries (10 000) searched against an index resident in
HBM; queries and result buffers are in HBM when the timed region starts; every step searches a DIFFERENT batch.
Configurations (BASELINE.json `configs`, generators of SURVEY.md 8d; no dataset can be downloaded here):
  c2 (default)  SIFT-1M stand-in S1: 1M x 128 float32, integer-valued, L2           ef by the recall rule (+ ef=100 line)
  c3            10M x 768 randn, rows normalised, inner product ("angular")        ef=200 as worded (recall is hopeless)
  c3-lowrank    10M x 768 S3 low-rank unit vectors, inner product                  ef by the recall rule
  c4            GloVe-1.2M stand-in: 1 183 514 x 100 low-rank unit vectors, IP     ef sweep 50..400, value at the recall rule
  c5            50M x 128 randn, L2, index replicated per GPU, queries sharded     ef=100 as worded (recall is hopeless)
  c5-lowrank    50M x 128 S1 generator (the SIFT stand-in at N=50M), L2            ef by the recall rule
  c2-uint8 / c5-uint8   the c2 / c5-lowrank data stored as bytes (integer datasets: bit-exact ids), L2   ef by the recall rule
All with M=32, ef_construction=100, K=10.  The metric's rule: the smallest ef of the sweep with recall@10 >= 0.95,
recall measured on all 10 000 queries of the first batch against exact brute force.

Output (round 5).  stdout carries exactly ONE line, printed last by rank 0: the contract line of the main configuration
(c2 by default), at most 4 KB -- contract fields, `roofline` (dominant kernel; achieved = algorithmic bytes per launch /
average launch duration from HIP events on the launch stream; `gather_ceiling` = what a pure random-row gather of this very
vector table reaches, measured in the run), `cpu_baseline` (N=1: the CPU oracle timed on the host cores),
`value_pcie_inclusive` (the same search through the host-buffer entry point: SURVEY 8d's definition of the metric),
`timed_regions` (value = the median of --regions timed regions of --steps launches each; min / max beside it) and
`secondary`: one short row per fixed-ef line and per FURTHER CONFIGURATION.  Those run after the main one in the same
process -- like the reference's harness, which runs a list of datasets and ef values in one invocation
(experiments/run-benchmark.py:362-506, tools/query_npy.cpp:132-158) -- N = 1: c2-uint8, c4, c3-lowrank, c3, c5, c5-lowrank, c5-uint8;
N > 1 GPUs: only c5, the configuration worded for 8 GPUs; `--secondary-configs none` turns them off, `--time-budget` skips
what no longer fits.  The FULL record -- every configuration's whole entry (own recall / ef rule, roofline, counters,
cpu_baseline with GPU ids == CPU ids, sustained and two-launches-in-flight rates, per-rank and broadcast reports) -- is
written to bench_out/bench_full.json (`--full-record`) and, as one line, to stderr.
Multi-GPU: one process per GPU; `--gpus N` without a torch.distributed environment spawns the N ranks itself.
Index replicated with one RCCL broadcast per buffer at load, queries sharded, no per-query collective; weak scaling.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20  # die-level L3 (same guide): part of a small index is served from it, and FETCH_SIZE counts those hits

CONFIGS = {
    # name: generator, n, dim, metric, dtype, fixed ef (0 = recall rule), sweep, fixed-ef secondary lines
    "c2": dict(gen="sift_like", n=1_000_000, dim=128, metric="l2", ef=0,
               sweep=[30, 40, 44, 48, 50, 52, 54, 56, 58, 60, 64, 70, 80, 100, 150, 200, 400], secondary=[100],
               title="SIFT-1M stand-in (S1 int-lowrank, SURVEY.md 8d)"),
    "c3": dict(gen="randn_unit", n=10_000_000, dim=768, metric="angular", ef=200, sweep=[], secondary=[],
               title="C3 as worded: randn rows normalised"),
    "c3-lowrank": dict(gen="lowrank_unit", n=10_000_000, dim=768, metric="angular", ef=0,
                       sweep=[100, 200, 300, 400, 600, 650, 660, 670, 680, 690, 700, 750, 800, 1000, 1200, 1600], secondary=[200],
                       title="C3 recall-qualified variant (S3 low-rank unit vectors, SURVEY.md 8d)"),
    "c4": dict(gen="glove_like", n=1_183_514, dim=100, metric="angular", ef=0,
               sweep=[50, 64, 80, 100, 104, 106, 108, 110, 120, 140, 170, 200, 400],
               secondary=[50, 100, 200, 400], title="GloVe-1.2M stand-in (rank-24 low-rank unit vectors, SURVEY.md 8d)"),
    "c5": dict(gen="randn", n=50_000_000, dim=128, metric="l2", ef=100, sweep=[], secondary=[],
               title="C5 as worded: randn, index replicated per GPU, queries sharded"),
    "c5-lowrank": dict(gen="sift_like", n=50_000_000, dim=128, metric="l2", ef=0,
                       sweep=[50, 64, 72, 76, 78, 80, 100, 128, 160, 200, 300, 400, 600], secondary=[],
                       title="C5 recall-qualified variant (the S1 SIFT stand-in generator at N=50M, SURVEY.md 8d)"),
    # north_star's bit-exact claim is about integer datasets: the c2 data stored as bytes (1-byte rows, v_dot4 arithmetic)
    "c2-uint8": dict(gen="sift_like", n=1_000_000, dim=128, metric="l2", ef=0, dtype="uint8",
                     sweep=[30, 40, 44, 48, 50, 52, 54, 56, 58, 60, 64, 70, 80, 100, 150, 200, 400], secondary=[],
                     title="SIFT-1M stand-in stored as uint8 (same integer-valued data as c2)"),
    # ... and at a size where HBM, not the Infinity Cache, is the roof (the reference's large sets are uint8:
    # experiments/data_loader.py:170-219 reads .u8bin): the c5-lowrank data stored as bytes, 6.4 GB of vectors + 6.4 GB of links
    "c5-uint8": dict(gen="sift_like", n=50_000_000, dim=128, metric="l2", ef=0, dtype="uint8",
                     sweep=[50, 64, 72, 76, 78, 80, 100, 128, 160, 200, 300, 400, 600], secondary=[],
                     title="the S1 SIFT stand-in generator at N=50M stored as uint8 (integer dataset at HBM scale)"),
}
# world size -> configurations after the main one (else: "c5")
SECONDARY_DEFAULT = {1: "c2-uint8,c4,c3-lowrank,c3,c5,c5-lowrank,c5-uint8"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--index-size", dest="n", type=int, default=0, help="override the configuration's node count")
    ap.add_argument("--nq", type=int, default=10_000, help="queries per batch per GPU")
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--efc", type=int, default=100)
    ap.add_argument("--ef", type=int, default=-1, help="ef_search; 0 = recall rule over --ef-sweep; default: the configuration's")
    ap.add_argument("--ef-sweep", default="")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--builder", default="device", choices=["device", "host"],
                    help="device: deterministic batched insertion on the GPU (same rule as Index::add); host: the "
                         "multi-threaded host builder (graph differs from run to run, like the reference's)")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--dtype", default="float32", choices=["float32", "uint8"],
                    help="index element type (uint8: c2 only -- the same integer-valued data stored as bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=1.0)
    ap.add_argument("--opt", action="append", default=[], help="C-ABI option name=value")
    ap.add_argument("--secondary-configs", default="default",
                    help="comma-separated configurations to run after the main one, each a full entry under its own "
                         "top-level key (default: c4,c3-lowrank,c5,c5-lowrank on one GPU, c5 on several; 'none' = off)")
    ap.add_argument("--secondary-index-size", type=int, default=0,
                    help="node count of the secondary configurations (0 = each configuration's own; tests use small ones)")
    ap.add_argument("--full-record", default="", help="where the full record goes (default: bench_out/bench_full.json)")
    ap.add_argument("--regions", type=int, default=3,
                    help="timed regions of --steps launches each per configuration; value = the median region's (min / max reported)")
    ap.add_argument("--time-budget", type=float, default=1500.0,
                    help="seconds after which no further secondary configuration is started")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------------------------------------------
class Data:
    """Base vectors in chunks (host numpy, the form Index.add takes) and query batches."""

    def __init__(self, cfg, n, nq_total, torch, device):
        self.cfg, self.n, self.dim = cfg, n, cfg["dim"]
        gen = cfg["gen"]
        self.on_device = n > 2_000_000  # big sets are generated by torch on the GPU, chunk by chunk
        self.torch, self.device = torch, device
        if not self.on_device:
            from flatnav_amd import datasets as ds

            if gen == "sift_like":
                self.X, self.Q = ds.sift_like(n, nq_total)
            elif gen == "lowrank_unit":
                self.X, self.Q = ds.lowrank_normalized(n, nq_total, dim=self.dim, rank=32, seed=7712)
            elif gen == "glove_like":
                self.X, self.Q = ds.lowrank_normalized(n, nq_total, dim=self.dim, rank=24, seed=100)
            elif gen == "randn_unit":
                self.X, self.Q = ds.randn(n, nq_total, self.dim, seed=768, normalize=True)
            else:
                self.X, self.Q = ds.randn(n, nq_total, self.dim, seed=50)
            self.note = "numpy generators of flatnav_amd/datasets.py (SURVEY.md 8d), base first, then queries"
        else:
            g = torch.Generator(device=device)
            seed = {"lowrank_unit": 7712, "glove_like": 100, "randn_unit": 768, "randn": 50, "sift_like": 1296}[gen]
            g.manual_seed(seed)
            self.g = g
            rank = 24 if gen == "glove_like" else 32
            self.W = None
            if gen in ("lowrank_unit", "glove_like"):
                self.W = torch.randn((rank, self.dim), generator=g, device=device) / (rank ** 0.5)
            if gen == "sift_like":
                self.W = torch.randn((16, self.dim), generator=g, device=device) / 4
            self.note = ("same distributions as SURVEY.md 8d, generated chunk-wise by torch on the GPU (seed %d): the "
                         "random stream differs from numpy's" % seed)
            self.nq_total = nq_total
            self.Q = None

    def _gen(self, m):
        torch, gen, g, dev = self.torch, self.cfg["gen"], self.g, self.device
        if gen in ("lowrank_unit", "glove_like"):
            x = torch.randn((m, self.W.shape[0]), generator=g, device=dev) @ self.W
            x += 0.05 * torch.randn((m, self.dim), generator=g, device=dev)
        elif gen == "sift_like":
            x = 64 + 32 * (torch.randn((m, 16), generator=g, device=dev) @ self.W)
            x += 6 * torch.randn((m, self.dim), generator=g, device=dev)
            return torch.clip(torch.round(x), 0, 255)
        else:
            x = torch.randn((m, self.dim), generator=g, device=dev)
        if gen != "randn":
            x /= x.norm(dim=1, keepdim=True)
        return x

    def chunks(self, rows):
        """Yields (first, host float32 array) over the base set."""
        if not self.on_device:
            for s in range(0, self.n, rows):
                yield s, self.X[s:s + rows]
        else:
            for s in range(0, self.n, rows):
                yield s, self._gen(min(rows, self.n - s)).cpu().numpy()

    def queries(self):
        if self.Q is None:
            self.Q = self._gen(self.nq_total).cpu().numpy()
        return self.Q


class Ctx:
    """What every configuration of one invocation shares: torch, the process group, this rank's GPU."""

    def __init__(self, torch, dist, rank, world, local_rank, dev_t, hw, t_start):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.local_rank, self.dev_t, self.hw, self.t_start = local_rank, dev_t, hw, t_start


def main() -> None:
    args = parse_args()
    t_start = time.time()
    # ---- N > 1 without a torch.distributed environment: spawn the ranks (before anything touches the GPU) -------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
        raise SystemExit(subprocess.call(cmd))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    from flatnav_amd import datasets as ds

    hw = ds.effective_cpus()  # honours the cgroup CPU quota (16 on the MI355X boxes, 256 CPUs visible)
    ctx = Ctx(torch, dist, rank, world, local_rank, torch.device("cuda", local_rank), hw, t_start)

    out = run_config(ctx, args, args.config, main_line=True)

    # ---- further configurations, each a full entry under its own key (module docstring) ---------------------------
    names = args.secondary_configs
    if names == "default":
        names = SECONDARY_DEFAULT.get(world, "c5")
        if args.n or args.ef >= 0 or args.dtype != "float32" or args.no_secondary or args.opt:  # a reduced / special run is about its one configuration
            names = "none"
    names = [] if names in ("none", "") else [n for n in names.split(",") if n and n != args.config]
    for name in names:
        if name not in CONFIGS:
            raise SystemExit("unknown configuration in --secondary-configs: %s" % name)
        # every rank must take the same decision: rank 0's clock decides
        go = (time.time() - t_start) < args.time_budget
        if dist is not None:
            flag = torch.tensor([1 if go else 0], device=ctx.dev_t)
            dist.broadcast(flag, src=0)
            go = bool(flag.item())
        if not go:
            if rank == 0:
                out[name] = {"skipped": "time budget of %.0f s used up (%.0f s elapsed)" % (args.time_budget, time.time() - t_start)}
                out["secondary"].append({"config": name, "skipped": True})
            continue
        sub = argparse.Namespace(**vars(args))
        sub.n, sub.ef, sub.ef_sweep, sub.opt = args.secondary_index_size, -1, "", list(args.opt)
        sub.steps, sub.warmup = max(3, min(args.steps, 10)), max(1, min(args.warmup, 3))
        sub.sustain_seconds = 0.0
        t0 = time.time()
        try:
            entry = run_config(ctx, sub, name, main_line=False)
        except Exception as exc:  # a further configuration must never take the contract line down with it
            if world > 1:
                raise  # (the other ranks are inside collectives: there is no recovering)
            import traceback

            traceback.print_exc()
            out[name] = {"skipped": "failed: %s: %s" % (type(exc).__name__, str(exc)[:300])}
            out["secondary"].append({"config": name, "skipped": True, "error": type(exc).__name__})
            torch.cuda.empty_cache()
            continue
        if rank == 0:
            entry["wall_seconds"] = round(time.time() - t0, 1)
            out[name] = entry
            out["secondary"].append({"config": name, "value": entry["value"], "unit": entry["unit"],
                                     "ef_search": entry["config"]["ef_search"], "recall_at_10": entry["config"]["recall_at_10"],
                                     "roofline_frac": entry["roofline"]["frac"], "full_entry": "top-level key \"%s\"" % name})
            log("[bench] " + summary_row(name, entry))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["bench_wall_seconds"] = round(time.time() - t_start, 1)
        rows = [summary_row(args.config, out)] + [summary_row(n, out[n]) for n in names if isinstance(out.get(n), dict)]
        out["summary"] = rows
        for r in rows:
            log("[bench] " + r)
        # Round 5: the FULL record (every configuration's whole entry: ~25 KB) goes to a file and to stderr; stdout carries
        # ONE line, last, of at most CONTRACT_LINE_MAX bytes -- the contract fields of the main configuration with `roofline`
        # and `cpu_baseline`, and one short row per further configuration (round 4's single 25 KB line was not parsed by
        # the driver).  The reference's harness likewise writes a small machine-readable metrics record
        # (experiments/run-benchmark.py:329-343).
        full_path = write_full_record(compact(out), args.full_record)
        print(contract_line(out, names, args.config, full_path), flush=True)


CONTRACT_LINE_MAX = 4096  # bytes; the driver parses the last stdout line, a 15 KB one still parsed, a 25 KB one did not


def write_full_record(full, path):
    """The whole record as pretty JSON: to `path` (default bench_out/bench_full.json under the repository, best effort --
    a read-only tree must not cost the contract line) and, as one line, to stderr."""
    log("[bench] full record: " + json.dumps(full, separators=(",", ":")))
    path = path or os.path.join(ROOT, "bench_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    except OSError as exc:
        log("[bench] could not write %s: %s" % (path, exc))
        return None


def _sig(x, digits=5):
    """Floats of the short rows: `digits` significant digits (the contract's own value / ms_per_step keep every digit)."""
    if isinstance(x, float) and x == x and abs(x) not in (0.0, float("inf")):
        return float("%.*g" % (digits, x))
    return x


def contract_line(out, names, main_name, full_path):
    """The ONE stdout line: contract fields + roofline + cpu_baseline of the main configuration, `value_pcie_inclusive`
    (SURVEY 8d defines the metric on the call that includes the copies), the three timed regions behind `value`, and one
    short row per further configuration.  Never longer than CONTRACT_LINE_MAX bytes: optional parts are dropped, in a
    fixed order, until it fits (a CPU test builds a seven-configuration record and checks the size)."""
    c, r = out["config"], out["roofline"]
    launch = c.get("launch") or {}
    traffic_rec = (r.get("traffic_recorded") or {}).get("hbm_bytes_per_launch_corrected")
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    #  which rate `value` is, in the line itself: the contract's device-resident rate; SURVEY 8d words the metric
    # on the call that includes the copies -- that number rides along as `value_pcie_inclusive`, named here
    what = "; value = device-resident rate (queries and results in HBM when the timed region starts)"
    if c.get("host_buffer_qps_pcie_inclusive"):
        what += ", SURVEY 8d's PCIe-inclusive rate (host buffers both ways) = %d queries/s = value_pcie_inclusive" % c["host_buffer_qps_pcie_inclusive"]
    line["config"] = {
        "workload": c["workload"][:320] + what,
        "ef_search": c["ef_search"],
        "recall_at_10": c["recall_at_10"],
        "recall_min_over_timed_batches": c["recall_all_timed_batches"]["min"],
        "timed_batches": c["recall_all_timed_batches"]["batches"],
        "parallelism": c["parallelism"],
        "kernel_variant": c.get("kernel_variant"),
        "exploratory_timed_launches": c.get("exploratory_timed_launches"),
        "queries_replayed_by_exact_kernel": c.get("queries_replayed_by_exact_kernel"),
        "queries_straight_to_exact_kernel": launch.get("tail_exact"),
        "launch": {k: launch.get(k) for k in ("kernel", "grid_blocks", "blocks_per_cu", "resident_per_cu", "lds_bytes", "visited_slots") if k in launch},
    }
    line["roofline"] = {
        "bound": r["bound"], "kernel": r["kernel"], "achieved": _sig(r["achieved"], 6), "peak": r["peak"], "unit": r["unit"],
        "frac": _sig(r["frac"], 6), "gather_ceiling": _sig(r.get("gather_ceiling"), 6),
        "frac_of_gather_ceiling": _sig(r.get("frac_of_gather_ceiling"), 4),
        "algorithmic_bytes_per_launch": _sig(r["algorithmic_bytes_per_launch"], 7), "avg_kernel_ms": _sig(r["avg_kernel_ms"], 6),
        "traffic": r.get("traffic"),
        "traffic_recorded": traffic_rec,
        "traffic_over_algorithmic": None if not traffic_rec else _sig(traffic_rec / r["algorithmic_bytes_per_launch"], 3),
    }
    cpu = out.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": _sig(cpu["value"], 6), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                "sample": cpu.get("sample_short") or cpu["sample"][:160]}
    line["value_pcie_inclusive"] = c.get("host_buffer_qps_pcie_inclusive")
    if c.get("host_buffer_qps_pinned_caller_zero_copy"):
        line["value_pcie_inclusive_pinned_caller"] = c["host_buffer_qps_pinned_caller_zero_copy"]  # (zero-copy: the caller's arrays are pinned)
    line["timed_regions"] = {k: (_sig(v) if not isinstance(v, list) else [_sig(x, 4) for x in v])
                             for k, v in (out.get("timed_regions") or {}).items() if k != "note"} or None
    if out.get("sustained"):  # >= 1 s of back-to-back steps: the timed region of the contract (--steps launches) is ~20 ms
        line["sustained"] = {"value": _sig(out["sustained"]["value"]), "seconds": _sig(out["sustained"]["seconds"], 3),
                             "steps": out["sustained"]["steps"]}
    if out.get("pipelined"):
        line["two_launches_in_flight"] = _sig(out["pipelined"]["value"])
    if out.get("single_query"):
        line["single_query_ms"] = {"ef": out["single_query"]["ef_search"], "wall_p50": _sig(out["single_query"]["wall_ms_p50"], 3),
                                   "kernel_p50": _sig(out["single_query"]["kernel_ms_p50"], 3)}
    mg = c.get("multi_gpu")
    if mg:  # three numbers; per-rank lists, the broadcast report and the peer matrix are in the full record
        rates = mg.get("per_rank_queries_per_s") or [0.0]
        gbps = [b.get("GBps") or 0.0 for b in (mg.get("index_broadcast") or [])]
        line["multi_gpu"] = {"slowest_rank_qps": _sig(min(rates)), "fastest_rank_qps": _sig(max(rates)),
                             "index_broadcast_GBps_min": _sig(min(gbps)) if gbps else None}
    sec = []
    for e2 in out.get("secondary", []):
        if "config" not in e2:  # a fixed-ef line of the main configuration
            sec.append({"config": main_name, "ef": e2["ef_search"], "recall": e2.get("recall_at_10"), "value": _sig(e2["value"]),
                        "frac": _sig(e2.get("roofline_frac"), 3)})
            continue
        e = out.get(e2["config"]) or {}
        if e2.get("skipped") or "skipped" in e:
            sec.append({"config": e2["config"], "skipped": str(e.get("skipped", e2.get("error", True)))[:80]})
            continue
        cpu2 = e.get("cpu_baseline")
        reg = e.get("timed_regions") or {}
        sec.append({"config": e2["config"], "ef": e["config"]["ef_search"], "recall": e["config"]["recall_at_10"],
                    "recall_min": e["config"]["recall_all_timed_batches"]["min"], "value": _sig(e["value"]),
                    "min": _sig(reg.get("min")), "max": _sig(reg.get("max")),
                    "frac": _sig(e["roofline"]["frac"], 3), "kernel_ms": _sig(e["roofline"]["avg_kernel_ms"], 4),
                    "pcie": e["config"].get("host_buffer_qps_pcie_inclusive"),
                    "cpu": None if not cpu2 else _sig(cpu2["value"], 4)})
    line["secondary"] = sec
    line["bench_wall_seconds"] = out.get("bench_wall_seconds")
    line["full_record"] = full_path
    # what may go, in this order, if a line ever grows past the limit
    droppable = [("roofline", "traffic_over_algorithmic"), ("config", "launch"), ("config", "parallelism"), ("cpu_baseline", "sample"),
                 (None, "value_pcie_inclusive_pinned_caller"), (None, "single_query_ms"), (None, "two_launches_in_flight"), (None, "timed_regions"), (None, "multi_gpu"), (None, "sustained")]
    text = json.dumps(line, separators=(",", ":"))
    while len(text.encode()) > CONTRACT_LINE_MAX:
        if droppable:
            parent, key = droppable.pop(0)
            (line if parent is None else line.get(parent, {})).pop(key, None)
        elif any(len(row) > 3 for row in line["secondary"]):
            for row in line["secondary"]:  # rows shrink to config / value / frac
                for k in [k for k in row if k not in ("config", "value", "frac", "skipped")]:
                    row.pop(k)
        elif line["secondary"]:
            line["secondary"].pop()
        else:
            line["config"]["workload"] = line["config"]["workload"][:100] + what[:70]
            text = json.dumps(line, separators=(",", ":"))
            break
        text = json.dumps(line, separators=(",", ":"))
    return text


def peer_matrix(world):
    """hipDeviceCanAccessPeer over the devices the ranks use (BENCH_SHARE_GPU: they all use device 0)."""
    from flatnav_amd import multigpu

    import torch

    return multigpu.peer_access_matrix(min(world, torch.cuda.device_count()))


def resident_per_cu(blocks_per_cu, lds_bytes):
    """Single-wave workgroups of `lds_bytes` bytes of LDS that one gfx950 CU really keeps resident, out of a grid of
    `blocks_per_cu` per CU: LDS is handed out in 1280-byte granules, 128 of them per CU (measured: tools/dev/probes/lds_granule.cpp,
    profiles/r4_launch_timeline.md section 2)."""
    granules = -(-max(int(lds_bytes), 1) // 1280)
    return min(int(blocks_per_cu), 128 // granules)


def summary_row(name, e):
    """'<config>: ef=.. recall=.. (min over batches ..) <queries/s> frac=.. of 8 TB/s (.. of the gather ceiling) cpu=..'"""
    if "skipped" in e:
        return "%s: skipped (%s)" % (name, e["skipped"])
    c, r = e["config"], e["roofline"]
    cpu = e.get("cpu_baseline")
    return ("%s: ef=%d recall@10=%.4f (min over %d batches %.4f) %.4g queries/s kernel %.4g ms frac=%.3f of 8 TB/s (%.2f of its gather "
            "ceiling) %d/CU (%d resident) cpu=%s"
            % (name, c["ef_search"], c["recall_at_10"], c["recall_all_timed_batches"]["batches"], c["recall_all_timed_batches"]["min"],
               e["value"], r["avg_kernel_ms"], r["frac"], r["frac_of_gather_ceiling"], c["launch"]["blocks_per_cu"],
               c["launch"].get("resident_per_cu", c["launch"]["blocks_per_cu"]),
               "-" if not cpu else "%.4g q/s on %d threads" % (cpu["value"], cpu["cores"])))


def compact(obj, top=True):
    """The line carries five configurations: floats below the top level are rounded to six significant digits (the
    contract fields keep every digit) so that it stays a line (~14 KB)."""
    if isinstance(obj, dict):
        return {k: (v if top and k in ("value", "ms_per_step") else compact(v, False)) for k, v in obj.items()}
    if isinstance(obj, list):
        return [compact(v, False) for v in obj]
    if isinstance(obj, float) and obj == obj and abs(obj) not in (0.0, float("inf")):
        return float("%.6g" % obj)
    return obj


def run_config(ctx, args, config, main_line):
    """Builds the index of one configuration, selects ef, times `args.steps` launches and returns (rank 0) the JSON
    object of that configuration -- the contract line when `main_line`, else a full entry of the same shape."""
    import gc

    import flatnav_amd as flatnav
    from flatnav_amd import hip

    torch, dist, rank, world, local_rank, dev_t, hw = ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.local_rank, ctx.dev_t, ctx.hw
    cfg = dict(CONFIGS[config])
    N = args.n or cfg["n"]
    NQ, DIM, M, K = args.nq, cfg["dim"], args.M, args.K
    metric = cfg["metric"]
    DT = cfg.get("dtype", args.dtype)
    if DT == "uint8" and cfg["gen"] != "sift_like":
        raise SystemExit("--dtype uint8 needs the integer-valued c2 data")
    ESIZE = 4 if DT == "float32" else 1

    # ---- data: `nb` distinct query batches per rank (one per step, reused cyclically beyond 32) -------------------
    nb = max(1, min(args.steps, 32))
    t0 = time.time()
    data = Data(cfg, N, NQ * nb * world, torch, dev_t)
    log("[rank %d] %s: data %.1fs (%s)" % (rank, config, time.time() - t0, data.note))

    # ---- index: rank 0 builds, the other ranks receive it by RCCL broadcast --------------------------------------
    index = None
    build_note = ""
    if rank == 0:
        t0 = time.time()
        index = flatnav.index.create(distance_type=metric, index_data_type=getattr(flatnav.data_type.DataType, DT),
                                     dim=DIM, dataset_size=N, max_edges_per_node=M)
        threads = args.build_threads or max(1, min(hw + hw // 2, os.cpu_count() or 1))
        index.set_num_threads(threads)
        index.set_device(local_rank)
        rows = 1_000_000 if DIM > 256 else 5_000_000
        for first, xh in data.chunks(rows):
            if DT == "uint8":
                xh = xh.astype(np.uint8)
            index.add(data=xh, ef_construction=args.efc, labels=list(range(first, first + len(xh))),
                      device=(args.builder == "device"))
            if data.on_device:
                log("[rank 0]   %d / %d nodes, %.1fs" % (first + len(xh), N, time.time() - t0))
        build_s = time.time() - t0
        build_note = ("device builder (fnv_index_insert_batch: batched insertion on the GPU, deterministic), %.1fs"
                      if args.builder == "device" else "host builder, %d threads, %%.1fs" % threads) % build_s
        log("[rank 0] build: %d nodes, %s" % (N, build_note))
        dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)  # the handle belongs to `index`
    else:
        dev = hip.DeviceIndex.alloc(M, N, DT, metric, DIM, device=local_rank)
    bcast = None
    if world > 1:
        from flatnav_amd import multigpu

        t0 = time.time()
        bcast = multigpu.replicate_index(dev, local_rank, src=0)  # RCCL broadcasts over xGMI, <= 2 GB pieces, at load only
        log("[rank %d] index broadcast %.2fs: %s" % (rank, time.time() - t0, ", ".join(
            "%s %.2f GB in %d pieces %.1f GB/s" % (b["buffer"], b["bytes"] / 1e9, b["pieces"], b["GBps"] or 0) for b in bcast)))
    for o in args.opt:
        k, v = o.split("=")
        dev.set_option(k, int(v))
    # Every NQ-query search this function asks for is counted, so that a kernel trace of this process can find the timed
    # region by position from the END of its full-grid launches (fnv_tune's own launches all come before it):
    # `roofline.trace_position` = {timed: launches of the timed region, after: launches that followed it}.
    import threading

    launches = {"n": 0, "timed_end": 0, "timed_open": False}
    launches_lock = threading.Lock()  # (the host entry point is also called from several threads below)
    _sd, _sh, _si = dev.search_device, dev.search, dev.search_into

    def _count_device(*a, **k):
        with launches_lock:
            launches["n"] += 1
        return _sd(*a, **k)

    def _count_host(*a, **k):
        with launches_lock:
            launches["n"] += 1
        return _sh(*a, **k)

    def _count_into(*a, **k):
        with launches_lock:
            launches["n"] += 1
        return _si(*a, **k)

    dev.search_device, dev.search, dev.search_into = _count_device, _count_host, _count_into
    ROW = dev.row_bytes  # bytes one row occupies in HBM (>= DIM * ESIZE: 16-byte chunks, whole 128-byte lines when cheap)
    TAIL = getattr(dev, "tail_bytes", 0)  # split rows (round 6): the row's last chunks live in a small cache-resident side table
    # bytes of the 128-byte lines one row touches: the stride itself when rows are whole lines, else the expectation for
    # a row that starts at a random 16-byte boundary inside a line (split rows: the main table's lines; the side table is
    # meant to be served from L2 / Infinity Cache)
    LINE_ROW = ROW if ROW % 128 == 0 else ROW + 112

    # ---- device-resident inputs / outputs -----------------------------------------------------------------------
    Q_all = data.queries()
    if DT == "uint8":
        Q_all = Q_all.astype(np.uint8)
    Q_rank = np.ascontiguousarray(Q_all[rank * NQ * nb:(rank + 1) * NQ * nb]).reshape(nb, NQ, DIM)
    dq = torch.from_numpy(Q_rank).to(dev_t)
    d_dist = torch.empty((NQ, K), dtype=torch.float32, device=dev_t)
    d_lab = torch.empty((NQ, K), dtype=torch.int32, device=dev_t)
    d_cnt = torch.empty(NQ, dtype=torch.int32, device=dev_t)
    d_nd = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
    d_nh = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
    stream = torch.cuda.current_stream()

    # ---- exact ground truth for recall@10: brute force on the GPU against the index's own HBM vector table,
    #      all NQ queries of this rank's first batch (rank 0 decides ef; every rank needs only ef) ----------------
    #      Round 4: EVERY batch the timed region searches has its ground truth, the rule has to hold on each of them.
    gts = None
    if rank == 0:
        t0 = time.time()
        gts = [exact_topk(torch, dev, dq[b], K, N, DIM, DT, metric) for b in range(nb)]
        log("[rank 0] exact ground truth for %d x %d queries: %.1fs" % (nb, NQ, time.time() - t0))

    def recall_at(ef, b=0):
        dev.search_device(dq[b].data_ptr(), NQ, K, ef, 100, d_dist.data_ptr(), d_lab.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        dev.status()
        return float((d_lab.long().unsqueeze(2) == gts[b].unsqueeze(1)).any(dim=2).float().mean().item())

    def recall_all(ef):
        """recall@10 of every batch of the timed region -> (min, mean, list)."""
        r = [recall_at(ef, b) for b in range(nb)]
        return min(r), float(np.mean(r)), r

    # ---- ef_search: fixed by the configuration, or the metric's rule (rank 0 decides) ----------------------------
    sweep_rec = {}
    sweep_min = {}  # ef -> lowest recall over the batches, for the efs whose batch 0 passed
    EF = cfg["ef"] if args.ef < 0 else args.ef
    sweep = [int(x) for x in args.ef_sweep.split(",")] if args.ef_sweep else cfg["sweep"]
    if EF == 0:
        for ef in sorted(sweep):
            ok = False
            if rank == 0:
                sweep_rec[ef] = round(recall_at(ef), 4)
                ok = sweep_rec[ef] >= 0.95
                if ok:  # the rule holds on batch 0: it has to hold on every batch that will be timed
                    sweep_min[ef] = round(recall_all(ef)[0], 4)
                    ok = sweep_min[ef] >= 0.95
            if dist is not None:
                flag = torch.tensor([1 if ok else 0], device=dev_t)
                dist.broadcast(flag, src=0)
                ok = bool(flag.item())
            if ok:
                EF = ef
                break
        if EF == 0:
            EF = max(sweep)
        log("[rank %d] ef sweep %s (lowest batch: %s) -> ef_search=%d" % (rank, sweep_rec, sweep_min, EF))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    tuned = {}
    per_rank = []  # seconds each rank measured for the most recent timed region (N > 1)

    def run(ef, steps, warmup, min_seconds=0.0):
        """Times `steps` launches (batch i mod nb each); returns (elapsed, kernel_ms list, steps done)."""
        def step(i):
            dev.search_device(dq[i % nb].data_ptr(), NQ, K, ef, 100, d_dist.data_ptr(), d_lab.data_ptr(),
                              d_cnt.data_ptr(), d_nd.data_ptr(), d_nh.data_ptr(), stream=stream.cuda_stream)
        if ef not in tuned:  # the library's per-beam-width kernel choice, settled in ONE explicit call (fnv_tune) on batch
            t0 = time.perf_counter()  # 0 -- set-up like a JIT's compilation, outside every timed region
            dev.tune(int(dq[0].data_ptr()), K, ef, 100, nq=NQ)
            tuned[ef] = time.perf_counter() - t0
        for i in range(warmup):
            step(i)
            torch.cuda.synchronize()  # untimed
        barrier()
        dev.status()
        evs = []
        explored = 0
        barrier()
        t0 = time.perf_counter()
        i = 0
        while True:
            for _ in range(steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                step(i)
                b.record(stream)
                evs.append((a, b))
                explored += 1 if dev.launch_info()["exploratory"] else 0
                i += 1
            if min_seconds <= 0:
                break
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= min_seconds:
                break
        barrier()
        elapsed = time.perf_counter() - t0
        if launches["timed_end"] is None or launches["timed_open"]:
            launches["timed_end"] = launches["n"]  # (the main measurement's last timed region ends here)
        dev.status()
        per_rank.clear()
        if dist is not None:
            mine = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank.extend(float(x.item()) for x in every)  # each rank's own clock around the same barriers
            elapsed = max(per_rank)
        return elapsed, [a.elapsed_time(b) for a, b in evs], i, explored

    step_nodes = max(1, N // 100)
    n_scan = (N + step_nodes - 1) // step_nodes

    def launch_bytes(ef, batches):
        """Algorithmic HBM bytes of one launch, averaged over the given batches, from the kernel's own
        per-query counters; also the bytes of the 128-byte lines those rows touch (LINE_ROW each), mean evaluations / hops."""
        tot, tot_rows, nds, nhs = 0, 0, [], []
        for b in batches:
            dev.search_device(dq[b].data_ptr(), NQ, K, ef, 100, d_dist.data_ptr(), d_lab.data_ptr(), d_cnt.data_ptr(),
                              d_nd.data_ptr(), d_nh.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            nd, nh = d_nd.cpu().numpy().astype(np.int64), d_nh.cpu().numpy().astype(np.int64)
            tot += int(((n_scan + nd) * DIM * ESIZE + nh * M * 4 + K * 4).sum())
            tot_rows += int(((n_scan + nd) * LINE_ROW + nh * M * 4 + K * 4).sum())
            nds.append(nd.mean())
            nhs.append(nh.mean())
        return tot / len(batches), tot_rows / len(batches), float(np.mean(nds)), float(np.mean(nhs))

    def measure(ef, steps, warmup, min_seconds=0.0, regions=1):
        """`regions` timed regions of `steps` launches each (each bracketed by its own barriers; the warm-up precedes the
        first); the numbers are the MEDIAN region's, `regions_qps` lists all of them (round 5: one region of 10-20 launches
        is 10-20 ms -- box-to-box and run-to-run differences of 5-10 % were larger than what a round gained)."""
        runs = []
        for g in range(max(1, regions)):
            elapsed, kms, done, explored = run(ef, steps, warmup if g == 0 else 0, min_seconds)
            runs.append((NQ * world * done / elapsed, elapsed, kms, done, explored, list(per_rank)))
        order = sorted(range(len(runs)), key=lambda j: runs[j][0])
        qps, elapsed, kms, done, explored, pr = runs[order[len(order) // 2]]
        per_rank[:] = pr
        used = sorted(set(i % nb for i in range(done)))
        byts, row_byts, nd_mean, nh_mean = launch_bytes(ef, used[:8])
        avg_kernel_s = float(np.mean(kms)) / 1e3
        return dict(elapsed=elapsed, steps=done, qps=qps, bytes=byts, row_bytes=row_byts, nd=nd_mean,
                    nh=nh_mean, kernel_ms=avg_kernel_s * 1e3, achieved=byts / avg_kernel_s / 1e9,
                    explored=sum(r[4] for r in runs), regions_qps=[r[0] for r in runs],
                    regions_kernel_ms=[float(np.mean(r[2])) for r in runs])

    launches["timed_end"], launches["timed_open"] = None, True
    main_m = measure(EF, args.steps, args.warmup, regions=args.regions)
    launches["timed_open"] = False
    main_per_rank = list(per_rank)
    out = None
    if rank == 0:
        rec_min, rec_mean, rec_list = recall_all(EF)
        recall = rec_list[0]
        dev.search_device(dq[(args.steps - 1) % nb].data_ptr(), NQ, K, EF, 100, d_dist.data_ptr(), d_lab.data_ptr(),
                          d_cnt.data_ptr(), d_nd.data_ptr(), d_nh.data_ptr(), stream=stream.cuda_stream)  # (geometry of a timed launch)
        torch.cuda.synchronize()
        geom = dev.launch_geometry()
        # blocks_per_cu = the grid's share per CU = (since round 5) the slots a CU really keeps resident: gfx950 hands LDS out in
        # 1280-byte granules (tools/dev/probes/lds_granule.cpp); rounds 1-4 launched the occupancy API's count, which can be one more
        geom["resident_per_cu"] = resident_per_cu(geom["blocks_per_cu"], geom["lds_bytes"])
        info = dev.launch_info()
        replay = dev.replayed_queries()
        ceiling = dev.gather_ceiling(geom["blocks_per_cu"])  # a pure gather of this very table, same load pattern
        # informational: the host-buffer entry point (pageable H2D of the queries + kernel + D2H of results)
        dev.search(Q_rank[0], K, EF)  # (first call: the pinned result slab and the device staging areas are allocated)
        host_ts = []
        for i in range(7):  # median of seven calls (the GPU has just idled through the recall computations: the first ones run slow)
            t0 = time.perf_counter()
            dev.search(Q_rank[(i + 1) % nb], K, EF)
            host_ts.append(time.perf_counter() - t0)
        host_qps = NQ / float(np.median(host_ts))
        # ... and for a caller whose arrays are pinned host memory (torch pin_memory): zero-copy at any batch size -- the kernel reads
        # the queries from and writes the results into the caller's own memory (round 6)
        qpin = torch.from_numpy(Q_rank[:min(nb, 8)]).pin_memory()
        dpin, lpin = torch.empty((NQ, K), dtype=torch.float32).pin_memory(), torch.empty((NQ, K), dtype=torch.int32).pin_memory()
        dev.search_into(qpin[0].numpy(), K, EF, dpin.numpy(), lpin.numpy())
        pin_ts = []
        for i in range(7):
            t0 = time.perf_counter()
            dev.search_into(qpin[(i + 1) % qpin.shape[0]].numpy(), K, EF, dpin.numpy(), lpin.numpy())
            pin_ts.append(time.perf_counter() - t0)
        host_pinned_qps = NQ / float(np.median(pin_ts))
        _, l_check = dev.search(Q_rank[7 % qpin.shape[0]], K, EF)  # (the last batch searched above)
        host_pinned_same = bool(np.array_equal(l_check, lpin.numpy()))
        del qpin, dpin, lpin
        # ... and with two and four caller threads on the one handle (concurrent callers run on the handle's hidden lanes:
        # their copies and launches overlap -- the reference's search is callable from several threads at once)
        def callers(T, per):
            def work(t):
                for i in range(per):
                    dev.search(Q_rank[(t + i) % nb], K, EF)
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return T * per * NQ / (time.perf_counter() - t0)
        callers(4, 2)  # (first contention creates the lanes and their launch plans: outside the timed regions)
        host2_qps = callers(2, 8)
        host4_qps = callers(4, 8)
        log("[rank 0] host-buffer (PCIe-inclusive) path: %.0f queries/s; pinned caller arrays (zero-copy): %.0f queries/s (ids equal: %s); "
            "two / four caller threads: %.0f / %.0f queries/s" % (host_qps, host_pinned_qps, host_pinned_same, host2_qps, host4_qps))
    # ---- two batches in flight (rank-local, informational): a second handle on the same HBM buffers (fnv_index_view),
    #      a second stream, launches alternate -- the drain of one launch (its last, slowest queries at falling
    #      occupancy) overlaps the start of the next.  This is the rate a server that always has the next batch ready
    #      sustains; the contract line above is one batch at a time on one stream.
    pipelined = None
    if world == 1 and not args.no_secondary:
        view = dev.view()
        view.tune(int(dq[0].data_ptr()), K, EF, 100, nq=NQ)
        # two streams of their own (round 5: with the process's default stream as one of the two, the main line's launches did
        # not overlap in two of two runs -- 9.7 M where tools/dev/pipelined_probe.py measures 12.6 M on the same library)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs2 = (torch.empty((NQ, K), dtype=torch.float32, device=dev_t), torch.empty((NQ, K), dtype=torch.int32, device=dev_t))
        torch.cuda.synchronize()
        lanes = [(dev, s1, d_dist, d_lab), (view, s2, outs2[0], outs2[1])]
        psteps = max(args.steps, 20) if main_line else max(args.steps, 6)

        def pipe_run(n):
            for i in range(n):
                h, st, od_, ol_ = lanes[i % 2]
                h.search_device(dq[i % nb].data_ptr(), NQ, K, EF, 100, od_.data_ptr(), ol_.data_ptr(), stream=st.cuda_stream)
        pipe_run(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe_run(psteps)
        torch.cuda.synchronize()
        pel = time.perf_counter() - t0
        pipelined = {"handles": 2, "streams": 2, "steps": psteps, "value": NQ * psteps / pel, "unit": "queries/s",
                     "ms_per_step": pel / psteps * 1e3,
                     "note": "informational: two 10 000-query launches in flight (fnv_index_view + a second stream, "
                             "launches alternate); NOT the contract value (one batch at a time on one stream)"}
        view.status()
        view.close()
        del view, outs2
    # ---- one query per call (rank-local, informational): the reference's published protocol is one search call per query
    #      (experiments/run-benchmark.py:66-82).  Host buffers in, host buffers out, one caller thread: wall time per call and
    #      the kernel's share of it (HIP events of the library), median of 200 calls at this configuration's ef.
    single = None
    if world == 1 and not args.no_secondary and main_line and rank == 0:
        qs = Q_rank[0]  # (`_sh`: the uncounted entry point -- these are not NQ-query launches, see `launches` above)
        for i in range(20):
            _sh(qs[i % NQ:i % NQ + 1], K, EF)
        walls, kerns = [], []
        for i in range(200):
            j = (20 + i) % NQ
            t0 = time.perf_counter()
            _sh(qs[j:j + 1], K, EF)
            walls.append(time.perf_counter() - t0)
            kerns.append(dev.last_kernel_ms())
        g1 = dev.launch_geometry()
        single = {"ef_search": EF, "calls": 200, "wall_ms_p50": float(np.percentile(walls, 50)) * 1e3,
                  "wall_ms_p99": float(np.percentile(walls, 99)) * 1e3, "kernel_ms_p50": float(np.percentile(kerns, 50)),
                  "value": 1.0 / float(np.mean(walls)), "unit": "queries/s", "lds_bytes_per_slot": g1["lds_bytes"],
                  "note": "informational: one query per fnv_search_batch call from one thread, host buffers both ways"}
        log("[rank 0] one query per call at ef=%d: wall p50 %.3f ms (p99 %.3f), kernel p50 %.3f ms, %.0f queries/s" % (
            EF, single["wall_ms_p50"], single["wall_ms_p99"], single["kernel_ms_p50"], single["value"]))
    # ---- fixed-ef lines of this configuration (all ranks take part: the timing barrier is collective) ----------------
    secondary = []
    sustained = None
    if not args.no_secondary:
        if args.sustain_seconds > 0:
            sm = measure(EF, max(args.steps, 10), 5, args.sustain_seconds)
            sustained = {"steps": sm["steps"], "seconds": sm["elapsed"], "value": sm["qps"], "unit": "queries/s",
                         "ms_per_step": sm["elapsed"] / sm["steps"] * 1e3,
                         "note": ">= %.1f s of back-to-back steps over %d rotating query batches (the contract line "
                                 "above times exactly --steps launches)" % (args.sustain_seconds, nb)}
        for ef2 in cfg["secondary"]:
            m2 = measure(ef2, max(5, min(args.steps, 20)), min(5, args.warmup))
            rec2 = recall_at(ef2) if rank == 0 else None  # (batch 0: a fixed-ef line, not a recall-rule point)
            secondary.append({"ef_search": ef2, "value": m2["qps"], "unit": "queries/s",
                              "recall_at_10": None if rec2 is None else round(rec2, 4),
                              "ms_per_step": m2["elapsed"] / m2["steps"] * 1e3, "steps": m2["steps"],
                              "roofline_frac": m2["achieved"] / HBM_PEAK_GBPS,
                              "achieved_GBps": m2["achieved"], "mean_dist_evals_per_query": m2["nd"]})
    if rank == 0:
        kname = {"two_heaps": "fnv_dev::beam_search_kernel",
                 "merged_beam_registers": "fnv_dev::beam_search_merged_kernel",
                 "merged_beam_lds": "fnv_dev::beam_search_merged_kernel"}[geom["kernel"]]
        index_bytes = N * (ROW + TAIL + 4 * M + 4)
        out = {
            "metric": "qps_at_recall10_ge_0.95",
            "value": main_m["qps"],
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if DT == "float32" else "u8",
            "data": "synthetic",
            "config": {
                "workload": "%s [%s]: %d x %d %s %s, M=%d, ef_construction=%d, ef_search=%d, K=%d, %d batched queries "
                            "per GPU per step (a different batch every step), index in HBM"
                            % (config, cfg["title"], N, DIM, DT, "L2" if metric == "l2" else "inner product", M,
                               args.efc, EF, K, NQ),
                "recall_at_10": round(recall, 4),
                "recall_queries": NQ,
                "recall_all_timed_batches": {"min": round(rec_min, 4), "mean": round(rec_mean, 4), "batches": nb,
                                             "queries_per_batch": NQ},
                "ef_search": EF,
                "ef_selection": ("fixed (configuration / --ef)" if not sweep_rec else
                                 "smallest ef of the sweep with recall@10 >= 0.95 on all %d queries of batch 0 AND of every "
                                 "other timed batch (SURVEY.md 8d); batch-0 recalls: %s; lowest batch at the efs that "
                                 "passed on batch 0: %s" % (NQ, sweep_rec, sweep_min)),
                "data_note": data.note,
                "index_build": build_note,
                "parallelism": "index replicated x%d, queries sharded" % world,
                "multi_gpu": None if world == 1 else {
                    "per_rank_queries_per_s": [NQ * args.steps / t for t in main_per_rank],
                    "per_rank_seconds": main_per_rank,
                    "index_broadcast": bcast,
                    "peer_access": peer_matrix(world),
                    "note": "value = all ranks' queries / the slowest rank's seconds; the index went out from rank 0 in "
                            "<= 2 GB RCCL broadcasts (per buffer: bytes, pieces, seconds, GB/s as rank 0 saw them)"},
                "mean_dist_evals_per_query": main_m["nd"],
                "mean_hops_per_query": main_m["nh"],
                "launch": geom,
                "kernel_variant": info["variant"],
                "kernel_choice": "fnv_tune: every variant measured on batch 0 in one explicit call before the warm-up "
                                 "(%.3f s); timed launches that were exploratory samples of the adaptive choice: %d"
                                 % (tuned.get(EF, 0.0), main_m["explored"]),
                "exploratory_timed_launches": main_m["explored"],
                "queries_replayed_by_exact_kernel": replay["total"],
                "host_buffer_qps_pcie_inclusive": round(host_qps),
                "host_buffer_qps_pinned_caller_zero_copy": round(host_pinned_qps),
                "host_buffer_pinned_caller_ids_equal": host_pinned_same,
                "host_buffer_qps_two_caller_threads": round(host2_qps),
                "host_buffer_qps_four_caller_threads": round(host4_qps),
                "index_bytes_in_hbm": index_bytes,
                "index_fraction_in_infinity_cache": round(min(1.0, INFINITY_CACHE_BYTES / index_bytes), 3),
                "measured_in_this_run": "value, ms_per_step, recall, roofline.achieved/avg_kernel_ms, gather ceiling, "
                                        "counters, secondary, sustained, cpu_baseline; roofline.traffic is null (PMC "
                                        "passes are separate rocprofv3 runs: see profiles/)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": main_m["achieved"],
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": main_m["achieved"] / HBM_PEAK_GBPS,
                "gather_ceiling": ceiling,
                "frac_of_gather_ceiling": main_m["achieved"] / ceiling,
                "gather_ceiling_note": "fnv_gather_ceiling, measured in this run: GB/s of algorithmic row bytes that a pure "
                                       "gather of random rows of THIS vector table (%d-byte rows at a %d-byte stride%s, "
                                       "%.2f GB) reaches with the search kernel's load pattern at %d waves per CU and no "
                                       "other work" % (DIM * ESIZE, ROW, " + %d bytes per row in the side table (not gathered here)" % TAIL if TAIL else "",
                                                       N * ROW / 1e9, geom["blocks_per_cu"]),
                "traffic": None,
                "traffic_recorded": recorded_traffic(config, DT, N, NQ, EF),
                "algorithmic_bytes_per_launch": main_m["bytes"],
                "row_bytes": DIM * ESIZE,
                "row_stride_bytes": ROW,
                "row_tail_bytes": TAIL,
                "line_bytes_per_launch": main_m["row_bytes"],
                "achieved_line_GBps": main_m["row_bytes"] / (main_m["kernel_ms"] / 1e3) / 1e9,
                "avg_kernel_ms": main_m["kernel_ms"],
                # the main measurement's timed regions are back to back in a kernel trace but for the launches that read
                # the counters after each of them; filled in below, once every launch has been made
                "trace_position": {"timed": args.steps, "regions": max(1, args.regions), "after": None},
            },
            "timed_regions": {"n": len(main_m["regions_qps"]), "min": min(main_m["regions_qps"]), "median": main_m["qps"],
                              "max": max(main_m["regions_qps"]), "kernel_ms": main_m["regions_kernel_ms"],
                              "note": "value / ms_per_step / roofline are the median region's (each region: --steps launches)"},
            "secondary": secondary,
            "sustained": sustained,
            "pipelined": pipelined,
            "single_query": single,
        }
        if not main_line:  # a further configuration's entry: the notes that repeat the main line's are dropped
            out.pop("sustained")
            out["ef_lines"] = out.pop("secondary")
            out["config"].pop("measured_in_this_run")
            out["roofline"].pop("gather_ceiling_note")
            if out.get("pipelined"):
                out["pipelined"].pop("note")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(index, dev, Q_rank[0], K, EF, hw, DT, metric, seconds=8.0 if main_line else 5.0)
            if not main_line:  # (the host is described once, in the main line's sample)
                out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"].split("; host:")[0]
    if out is not None:
        out["roofline"]["trace_position"]["after"] = launches["n"] - launches["timed_end"]
    # ---- give everything back before the next configuration ----------------------------------------------------------
    dev.search_device, dev.search, dev.search_into = _sd, _sh, _si
    dev.close()
    del dev, index, dq, d_dist, d_lab, d_cnt, d_nd, d_nh, gts, data, Q_all, Q_rank
    gc.collect()
    torch.cuda.empty_cache()
    return out


def exact_topk(torch, dev, q, K, N, DIM, DT, metric, block=250_000):
    """Exact top-K node labels (== row numbers here) of every query in `q` by brute force against the index's HBM
    vector table: fp32 GEMM scores per block of rows, exact re-ranking of each block's best 4K candidates."""
    from flatnav_amd import multigpu

    (vptr, vbytes), _, _ = dev.device_buffers()
    row_bytes = dev.row_bytes
    tail_bytes = getattr(dev, "tail_bytes", 0)
    dev_t = q.device
    table = torch.as_tensor(multigpu._DevView(vptr, N * row_bytes), device=dev_t)
    tails = None
    if tail_bytes:  # split rows: the side table follows the main table of the handle's capacity
        capacity = vbytes // (row_bytes + tail_bytes)
        tails = torch.as_tensor(multigpu._DevView(vptr + capacity * row_bytes, N * tail_bytes), device=dev_t)
    qf = q.float()
    best_s = torch.full((q.shape[0], K), float("inf"), device=dev_t)
    best_i = torch.zeros((q.shape[0], K), dtype=torch.int64, device=dev_t)
    for s in range(0, N, block):
        e = min(N, s + block)
        rows = table[s * row_bytes:e * row_bytes].view(e - s, row_bytes)
        if tails is not None:
            rows = torch.cat([rows, tails[s * tail_bytes:e * tail_bytes].view(e - s, tail_bytes)], dim=1)
        if DT == "float32":
            x = rows.contiguous().view(torch.float32)[:, :DIM]
        else:
            x = rows[:, :DIM].float()
        for qs in range(0, qf.shape[0], 2500):
            qq = qf[qs:qs + 2500]
            if metric == "l2":
                sc = (x * x).sum(1)[None, :] - 2.0 * (qq @ x.T)
            else:
                sc = -(qq @ x.T)
            cs, ci = torch.topk(sc, min(4 * K, e - s), dim=1, largest=False)
            # exact scores of the shortlisted rows (the GEMM form of L2 loses digits on large norms)
            xs = x[ci]  # [q, 4K, dim]
            ex = ((xs - qq[:, None, :]) ** 2).sum(2) if metric == "l2" else 1.0 - (xs * qq[:, None, :]).sum(2)
            alls = torch.cat([best_s[qs:qs + 2500], ex], 1)
            alli = torch.cat([best_i[qs:qs + 2500], ci + s], 1)
            o = torch.topk(alls, K, dim=1, largest=False).indices
            best_s[qs:qs + 2500] = alls.gather(1, o)
            best_i[qs:qs + 2500] = alli.gather(1, o)
        del x
    return best_i


def recorded_traffic(config, dtype, n, nq, ef):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, corrected as MI355X_MICROARCH.md prescribes) -- a RECORDED number, labelled as such; None unless
    the committed passes profiled this very workload."""
    if config == "c2-uint8":
        config = "c2"  # (the 1M uint8 index is recorded as config c2, dtype uint8)
    for name in ("r6_pmc_hbm_traffic.json", "r5_pmc_hbm_traffic.json", "r4_pmc_hbm_traffic.json", "r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json", "pmc_hbm_traffic.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        for r in rec if isinstance(rec, list) else [rec]:
            if (r.get("config", "c2"), r.get("dtype", "float32"), r.get("n"), r.get("nq"), r.get("ef")) == (config, dtype, n, nq, ef):
                return {"hbm_bytes_per_launch_corrected": r.get("hbm_bytes_per_launch_corrected"),
                        "source": "profiles/%s (recorded in an earlier rocprofv3 --pmc run, not in this run)" % name}
    return None


def cpu_baseline(index, dev, Q, K, EF, hw, dtype, metric, seconds=8.0):
    """The CPU oracle (restated reference search, oracle/) on the same graph and a bounded sample of the same queries,
    all usable host threads, roughly 10-20 s.  Also re-checks GPU == CPU ids on the sample."""
    from oracle import oracle as orc

    orc.build()
    blob = np.asarray(index._raw_blob())
    n = int(index._cur_num_nodes)
    o = orc.OracleIndex.from_blob(metric, dtype, Q.shape[1], n, n, index.max_edges_per_node, blob)
    kind_note = "oracle port, own AVX2 distance"
    if o.use_reference_distance(True):
        kind_note = "oracle port of Index::search driving the reference's own compiled AVX-512 distance kernel (oracle/_ref)"
    threads = hw
    t0 = time.perf_counter()
    o.search(Q[:256], K, EF, threads=threads)  # warm; also sizes the sample
    per_q = (time.perf_counter() - t0) / 256
    sample = int(min(len(Q), max(256, seconds / 2 / max(per_q, 1e-9))))
    reps, t_used, nq_done = 0, 0.0, 0
    ol = None
    while t_used < seconds and reps < 50:
        t0 = time.perf_counter()
        _, ol = o.search(Q[:sample], K, EF, threads=threads)
        t_used += time.perf_counter() - t0
        nq_done += sample
        reps += 1
    n1 = int(min(sample, max(64, 2.0 / max(per_q * threads, 1e-9))))
    t0 = time.perf_counter()
    o.search(Q[:n1], K, EF, threads=1)
    qps1 = n1 / (time.perf_counter() - t0)
    _, gl = dev.search(Q[:sample], K, EF)
    same = float((ol == gl).all(axis=1).mean())
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
        "sample_short": "%d x first %d queries of batch 0, %d threads (1 thread: %.0f q/s); GPU ids == CPU ids on %.2f%% of them; %s"
                        % (reps, sample, threads, qps1, same * 100, "reference's AVX-512 distance kernel" if "oracle/_ref" in kind_note else "own AVX2 distance"),
        "sample": "%d x the first %d queries of batch 0 on %d host threads (%s); single-thread: %.0f queries/s; "
                  "GPU ids == CPU ids on %.2f%% of the sample; host: %s, %d CPUs visible, %d usable (cgroup quota)"
                  % (reps, sample, threads, kind_note, qps1, same * 100, cpu_model, os.cpu_count() or 0, hw),
    }


if __name__ == "__main__":
    main()
