"""bench.py contract on the GPU box: the single-GPU line carries every required field, and the multi-rank
control flow (replication broadcast, ef-sweep agreement, barriers, max-over-ranks timing) runs with 2 ranks
sharing the one GPU over gloo (BENCH_SHARE_GPU=1; the real N>1 run is one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--index-size", "60000", "--nq", "2000", "--steps", "2", "--warmup", "1"]


def _last_json(out):
    """stdout carries exactly ONE line, the contract line, at most 4 KB (BENCH_r04.json: a 25 KB line was not parsed)."""
    lines = [l for l in out.splitlines() if l.strip()]
    # (the gloo backend of the two-rank tests prints "[Gloo] Rank ... is connected to ..." lines of its own to stdout, the two
    #  ranks' interleaved character by character -- fragments like a lone "1" come out as lines of their own; RCCL prints
    #  nothing) -- what the driver relies on: bench.py's own output is ONE line, it parses, and it is the LAST one.  What a
    #  backend scribbles before it is not asserted on (round 6: a fragment that did not contain "connected to" failed this
    #  check on one box and, under -x, hid the rest of the suite).
    ours = [l for l in lines if l.startswith("{")]
    assert len(ours) == 1 and lines[-1] == ours[0], out[-3000:]
    assert len(ours[0].encode()) <= 4096, len(ours[0])
    return json.loads(ours[0])


def _run(cmd, tmp_path, env=None, timeout=900):
    """-> (contract line, full record)"""
    full = str(tmp_path / "bench_full.json")
    out = subprocess.run(cmd + ["--full-record", full], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["full_record"] == full
    return d, json.load(open(full))


def test_single_gpu_line_has_the_contract_fields(tmp_path):
    d, full = _run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, tmp_path)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_pcie_inclusive", "timed_regions"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["config"]["recall_at_10"] >= 0.95 and "workload" in d["config"] and d["config"]["recall_min_over_timed_batches"] >= 0.95
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5  # (floats below the top level carry six digits)
    # (measured rates: present and positive -- that a pure gather beats gather + search is what the numbers show, by 4 x at
    #  this size, but a result-pinning suite run under -x asserts no inequality between two measurements)
    assert r["gather_ceiling"] > 0 and r["achieved"] > 0 and r["frac_of_gather_ceiling"] > 0
    assert r["algorithmic_bytes_per_launch"] > 0 and r["avg_kernel_ms"] > 0
    assert full["roofline"]["algorithmic_bytes_per_launch"] <= full["roofline"]["line_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "100.00%" in c["sample"]  # GPU ids == CPU ids
    assert d["config"]["exploratory_timed_launches"] == 0
    # value = the median of three timed regions of --steps launches each; the same number in line and file
    t = d["timed_regions"]
    assert t["n"] == 3 and t["min"] <= t["median"] <= t["max"] and abs(t["median"] - d["value"]) < 1e-4 * d["value"]
    assert full["value"] == d["value"] and full["ms_per_step"] == d["ms_per_step"] and d["value_pcie_inclusive"] > 0


def test_default_run_carries_the_other_configurations(tmp_path):
    # the default invocation runs the further BASELINE configurations after the main one: each is a full entry under its
    # own key of the full record and ONE short row of the contract line (here at test sizes: --secondary-index-size)
    d, full = _run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL +
                   ["--secondary-configs", "c2-uint8,c4,c5-lowrank", "--secondary-index-size", "50000"], tmp_path, timeout=1500)
    assert d["config"]["workload"].startswith("c2 ")
    rows = {x["config"]: x for x in d["secondary"] if x["config"] != "c2"}
    for name in ("c2-uint8", "c4", "c5-lowrank"):
        e = full[name]
        assert e["config"]["workload"].startswith(name + " ") and e["value"] > 0 and e["config"]["recall_at_10"] >= 0.95
        assert e["roofline"]["algorithmic_bytes_per_launch"] > 0 and e["roofline"]["frac"] > 0
        assert e["cpu_baseline"]["value"] > 0 and "GPU ids == CPU ids" in e["cpu_baseline"]["sample"]
        row = rows[name]
        assert abs(row["value"] - e["value"]) < 1e-3 * e["value"] and row["ef"] == e["config"]["ef_search"] and row["cpu"] > 0
        assert row["min"] <= row["value"] * 1.0001 and row["value"] <= row["max"] * 1.0001 and row["frac"] > 0  # (value = the median region)
    assert len(full["c4"]["ef_lines"]) == 4  # the fixed-ef sweep of c4
    assert len(full["summary"]) == 4


def test_two_ranks_share_one_gpu_over_gloo(tmp_path):
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    d, full = _run(cmd, tmp_path, env=env)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["config"]["parallelism"].startswith("index replicated x2")
    assert d["config"]["recall_at_10"] >= 0.95 and d["value"] > 0
    # three numbers in the line; per-rank rates, the broadcast report and the peer matrix in the file
    assert d["multi_gpu"]["slowest_rank_qps"] <= d["multi_gpu"]["fastest_rank_qps"] and d["multi_gpu"]["index_broadcast_GBps_min"] > 0
    mg = full["config"]["multi_gpu"]
    assert len(mg["per_rank_queries_per_s"]) == 2 and len(mg["index_broadcast"]) == 3 and len(mg["peer_access"]) >= 1


def test_two_ranks_also_run_the_multi_gpu_configuration(tmp_path):
    # with N > 1 GPUs the default line carries c5 (the configuration worded for 8 GPUs) after the main one: every rank
    # takes part in its broadcast, ef agreement and timing barriers (here at a test size, two ranks on the one GPU)
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + [
               "--secondary-configs", "c5", "--secondary-index-size", "40000"]
    d, full = _run(cmd, tmp_path, env=env)
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("c2 ")
    e = full["c5"]
    assert e["n_gpus"] == 2 and e["config"]["workload"].startswith("c5 ") and e["value"] > 0
    assert e["config"]["parallelism"].startswith("index replicated x2") and "cpu_baseline" not in e
    assert any(x.get("config") == "c5" for x in d["secondary"])


def test_benchmark_harness_writes_the_reference_metrics(tmp_path):
    # tools/run_benchmark.py: the reference's harness metrics (experiments/run-benchmark.py:38-124: recall, qps,
    # latency percentiles, distance computations per query) for the GPU index, written as metrics.json
    out_file = str(tmp_path / "metrics.json")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_benchmark.py"), "--synthetic", "sift", "--n", "20000",
           "--num-queries", "300", "--k", "10", "--ef-search", "50", "100", "--single-query-samples", "50",
           "--metrics-file", out_file, "--dataset-name", "sift-like-20k"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    exps = json.load(open(out_file))["sift-like-20k"]
    assert len(exps) == 2
    for e in exps:
        for k in ("recall", "qps", "qps_batched", "latency_p50", "latency_p95", "latency_p99", "latency_p999",
                  "distance_computations", "build_time", "index_size", "node_links", "ef_construction", "ef_search", "k"):
            assert k in e, k
        assert e["recall"] > 0.9 and e["qps"] > 0 and e["distance_computations"] > 100
    assert exps[1]["recall"] >= exps[0]["recall"] - 0.005  # (the wider beam; recall is not a theorem-grade monotone function of ef)


def test_bench_spawns_its_own_ranks_and_runs_other_configs(tmp_path):
    # `python bench.py --gpus 2` without a torch.distributed environment launches the two ranks itself; c4 (100-d
    # inner product) exercises a configuration other than the default, at a reduced size
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4", "--index-size", "50000",
           "--nq", "1000", "--steps", "2", "--warmup", "1", "--no-secondary"]
    d, full = _run(cmd, tmp_path, env=env)
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("c4 ") and d["value"] > 0
    assert d["roofline"]["frac_of_gather_ceiling"] > 0 and d["roofline"]["frac"] > 0  # (a 25 MB table gathers from L2: ceiling > HBM peak)
    # 100-d rows: three whole lines in the table + 16 bytes in the side table (split rows, round 6; rounds 2-5: a 512-byte stride)
    assert full["roofline"]["row_bytes"] == 400 and full["roofline"]["row_stride_bytes"] == 384 and full["roofline"]["row_tail_bytes"] == 16


@pytest.mark.parametrize("metric", ["l2", "angular"])
def test_bench_ground_truth_equals_numpy_brute_force(metric):
    # bench.py's recall is measured against exact_topk() (GEMM shortlist + exact re-rank on the index's own HBM vector
    # table); the float64 numpy brute force of flatnav_amd/datasets.py must give the same neighbour sets.
    import ctypes

    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    import bench
    import flatnav_amd as flatnav
    from flatnav_amd import datasets as ds, hip

    N, NQ, K = 30000, 700, 10
    if metric == "l2":
        X, Q = ds.sift_like(N, NQ)
        truth = ds.exact_topk_l2(X, Q, K)
    else:
        X, Q = ds.lowrank_normalized(N, NQ, dim=100, rank=24, seed=100)
        truth = ds.exact_topk_ip(X, Q, K)
    ix = flatnav.index.create(metric, X.shape[1], N, 16)
    ix.set_num_threads(4)
    ix.add(X, 40, device=True)
    dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
    got = bench.exact_topk(torch, dev, torch.from_numpy(Q).cuda(), K, N, X.shape[1], "float32",
                           "l2" if metric == "l2" else "ip", block=7000).cpu().numpy()
    same = np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got, truth)]) / K
    assert same > 0.9995  # equal distances at the K-th place may resolve differently; everything else must agree
