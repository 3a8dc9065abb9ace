"""bench.py's host-side helpers on the CPU (no GPU, no library): the summary rows that end the JSON line, the rounding of
the line, the configuration table.  A typo here would only show at the end of a ten-minute GPU run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _entry(value=1.5e6, cpu=True):
    e = {"value": value, "unit": "queries/s",
         "config": {"ef_search": 52, "recall_at_10": 0.9511, "launch": {"blocks_per_cu": 16},
                    "recall_all_timed_batches": {"min": 0.9507, "mean": 0.952, "batches": 20, "queries_per_batch": 10000}},
         "roofline": {"frac": 0.7791234567, "frac_of_gather_ceiling": 0.87, "avg_kernel_ms": 1.03}}
    if cpu:
        e["cpu_baseline"] = {"value": 143630.0, "cores": 16}
    return e


def test_summary_rows_name_every_configuration():
    row = bench.summary_row("c2", _entry())
    assert row.startswith("c2: ef=52 recall@10=0.9511 (min over 20 batches 0.9507)") and "16/CU" in row and "16 threads" in row
    assert "cpu=-" in bench.summary_row("c5", _entry(cpu=False))
    assert bench.summary_row("c3", {"skipped": "time budget used up"}) == "c3: skipped (time budget used up)"
    assert "failed: MemoryError" in bench.summary_row("c5", {"skipped": "failed: MemoryError: x"})


def test_compact_keeps_the_contract_fields_and_rounds_the_rest():
    line = {"value": 9634766.123456789, "ms_per_step": 1.0378912345, "roofline": {"frac": 0.7791234567, "peak": 8000.0},
            "c4": _entry(), "summary": ["c2: ...", "c4: ..."], "vs_baseline": None}
    out = bench.compact(line)
    assert out["value"] == line["value"] and out["ms_per_step"] == line["ms_per_step"]
    assert out["roofline"]["frac"] == 0.779123 and out["c4"]["value"] == 1.5e6 and out["vs_baseline"] is None
    assert list(out)[-2] == "summary"  # key order is kept: the summary stays where main() put it
    json.dumps(out)


def _full_entry(name, cpu=True, world=1):
    """An entry of the shape run_config() returns, with strings as long as the real ones."""
    e = {"metric": "qps_at_recall10_ge_0.95", "value": 9634766.123456789, "unit": "queries/s", "n_gpus": world, "steps": 20,
         "warmup": 5, "ms_per_step": 1.0378912345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
         "dtype": "f32", "data": "synthetic",
         "config": {"workload": "%s [%s]: 50000000 x 128 float32 inner product, M=32, ef_construction=100, ef_search=1600, K=10, "
                                "10000 batched queries per GPU per step (a different batch every step), index in HBM" % (name, "t" * 90),
                    "recall_at_10": 0.9511, "recall_queries": 10000, "ef_search": 1600,
                    "recall_all_timed_batches": {"min": 0.9507, "mean": 0.9521, "batches": 20, "queries_per_batch": 10000},
                    "ef_selection": "x" * 600, "data_note": "y" * 200, "index_build": "z" * 120,
                    "parallelism": "index replicated x%d, queries sharded" % world,
                    "multi_gpu": None if world == 1 else {
                        "per_rank_queries_per_s": [1.2e6 + i for i in range(world)], "per_rank_seconds": [0.1] * world,
                        "index_broadcast": [{"buffer": b, "bytes": 3.2e10, "pieces": 16, "seconds": 0.3, "GBps": 101.123456}
                                            for b in ("vectors", "links", "labels")],
                        "peer_access": [[1] * world] * world, "note": "n" * 200},
                    "launch": {"grid_blocks": 4096, "block_threads": 64, "lds_bytes": 10144, "blocks_per_cu": 16, "visited_slots": 2048,
                               "cand_slots": 296, "kernel": "merged_beam_registers", "tail_exact": 4096, "resident_per_cu": 16},
                    "kernel_variant": 5, "kernel_choice": "k" * 200, "exploratory_timed_launches": 0,
                    "queries_replayed_by_exact_kernel": 141, "host_buffer_qps_pcie_inclusive": 8234567,
                    "host_buffer_qps_two_caller_threads": 9400000, "host_buffer_qps_four_caller_threads": 11500000,
                    "index_bytes_in_hbm": 644000000, "index_fraction_in_infinity_cache": 0.417, "measured_in_this_run": "m" * 200},
         "roofline": {"bound": "hbm", "kernel": "fnv_dev::beam_search_merged_kernel", "achieved": 6228.123456789, "peak": 8000.0,
                      "unit": "GB/s", "frac": 0.7785154320986, "gather_ceiling": 7113.987654321, "frac_of_gather_ceiling": 0.87548,
                      "gather_ceiling_note": "g" * 300, "traffic": None,
                      "traffic_recorded": {"hbm_bytes_per_launch_corrected": 5838123456.0, "source": "profiles/r4_pmc_hbm_traffic.json (...)"},
                      "algorithmic_bytes_per_launch": 6415712345.5, "row_bytes": 512, "row_stride_bytes": 512,
                      "line_bytes_per_launch": 6415712345.5, "achieved_line_GBps": 6228.1, "avg_kernel_ms": 1.0301234,
                      "trace_position": {"timed": 20, "regions": 3, "after": 91}},
         "timed_regions": {"n": 3, "min": 9534766.1, "median": 9634766.123456789, "max": 9734766.9, "kernel_ms": [1.03, 1.031, 1.029], "note": "n" * 90},
         "secondary": [], "sustained": {"value": 9.5e6, "seconds": 1.0123, "steps": 970, "unit": "queries/s", "note": "n" * 100}, "pipelined": {"value": 12034567.8, "note": "p" * 150},
         "single_query": {"ef_search": 1600, "calls": 200, "wall_ms_p50": 0.17234567, "wall_ms_p99": 0.189, "kernel_ms_p50": 0.13912345,
                          "value": 5780.123, "unit": "queries/s", "lds_bytes_per_slot": 130160, "note": "s" * 100}}
    if cpu:
        e["cpu_baseline"] = {"value": 143630.123, "unit": "queries/s", "cores": 16, "kind": "port", "sample": "s" * 420,
                             "sample_short": "9 x first 10000 queries of batch 0, 16 threads (1 thread: 10912 q/s); GPU ids == CPU ids on "
                                             "100.00% of them; reference's AVX-512 distance kernel"}
    return e


def _record(names, world=1):
    out = _full_entry("c2", cpu=world == 1, world=world)
    out["secondary"] = [{"ef_search": 100, "value": 5.4e6, "unit": "queries/s", "recall_at_10": 0.9901, "roofline_frac": 0.7312345}]
    for n in names:
        e = _full_entry(n, cpu=world == 1, world=world)
        e["ef_lines"] = e.pop("secondary")
        out[n] = e
        out["secondary"].append({"config": n, "value": e["value"], "unit": "queries/s", "ef_search": 1600, "recall_at_10": 0.9511,
                                 "roofline_frac": 0.61, "full_entry": "top-level key"})
    out["bench_wall_seconds"] = 663.1
    return out


def test_contract_line_of_an_eight_configuration_run_stays_under_4_kb():
    # BENCH_r04.json: "parsed": null -- the one stdout line had grown to 25 KB.  Whatever the run holds, the line that goes to
    # stdout is at most 4 KB, parses, and carries roofline + cpu_baseline + one row per further configuration.
    names = bench.SECONDARY_DEFAULT[1].split(",")
    assert len(names) == 7 and names[-1] == "c5-uint8"  # (the last one: the first to be skipped if the time budget runs out)
    out = _record(names)
    out["c5"] = {"skipped": "failed: MemoryError: " + "x" * 300}
    out["secondary"][-3] = {"config": "c5", "skipped": True, "error": "MemoryError"}
    text = bench.contract_line(out, names, "c2", "bench_out/bench_full.json")
    assert len(text.encode()) < bench.CONTRACT_LINE_MAX and "\n" not in text
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "value_pcie_inclusive", "secondary"):
        assert k in d, k
    assert d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"]  # the contract's own numbers keep every digit
    assert d["config"]["workload"].startswith("c2 ") and d["config"]["ef_search"] == 1600 and d["config"]["recall_min_over_timed_batches"] == 0.9507
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["traffic"] is None
    assert abs(r["traffic_over_algorithmic"] - 0.91) < 0.005 and r["avg_kernel_ms"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16 and "100.00%" in d["cpu_baseline"]["sample"]
    assert d["value_pcie_inclusive"] == 8234567 and d["timed_regions"]["n"] == 3
    #  the line itself says which rate `value` is and where SURVEY 8d's PCIe-inclusive one stands; the >= 1 s figure rides along
    assert "value = device-resident rate" in d["config"]["workload"] and "8234567 queries/s = value_pcie_inclusive" in d["config"]["workload"]
    assert d["sustained"] == {"value": 9.5e6, "seconds": 1.01, "steps": 970}
    assert d["single_query_ms"] == {"ef": 1600, "wall_p50": 0.172, "kernel_p50": 0.139}  # (one query per call: the reference's own protocol)
    rows = {row["config"]: row for row in d["secondary"] if row["config"] != "c2"}
    assert set(rows) == set(names) and "skipped" in rows["c5"] and rows["c4"]["frac"] == 0.779 and rows["c4"]["cpu"] > 0
    assert all(len(json.dumps(row)) <= 220 for row in d["secondary"])
    # the 8-GPU shape: per-rank lists and the broadcast report stay in the file, three numbers in the line
    text8 = bench.contract_line(_record(["c5"], world=8), ["c5"], "c2", None)
    d8 = json.loads(text8)
    assert len(text8.encode()) < bench.CONTRACT_LINE_MAX and d8["n_gpus"] == 8 and "cpu_baseline" not in d8
    assert set(d8["multi_gpu"]) == {"slowest_rank_qps", "fastest_rank_qps", "index_broadcast_GBps_min"}


def test_contract_line_sheds_optional_parts_before_it_outgrows_the_limit():
    names = bench.SECONDARY_DEFAULT[1].split(",") * 6  # 36 further configurations: more rows than 4 KB hold
    out = _record(list(dict.fromkeys(names)))
    for i in range(30):
        out["secondary"].append(dict(out["secondary"][1], config="c4"))
    text = bench.contract_line(out, names, "c2", "bench_out/bench_full.json")
    d = json.loads(text)
    assert len(text.encode()) <= bench.CONTRACT_LINE_MAX
    assert d["value"] == out["value"] and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["secondary"]


def test_full_record_goes_to_a_file(tmp_path, capsys):
    path = str(tmp_path / "sub" / "bench_full.json")
    assert bench.write_full_record(bench.compact(_record(["c4"])), path) == path
    assert json.load(open(path))["c4"]["roofline"]["frac"] == 0.778515
    assert "[bench] full record: {" in capsys.readouterr().err
    assert bench.write_full_record({"a": 1}, "/proc/nonexistent/x.json") is None  # best effort: the line still goes out


def test_configuration_table_is_consistent():
    assert set(bench.SECONDARY_DEFAULT[1].split(",")) <= set(bench.CONFIGS) and "c2" not in bench.SECONDARY_DEFAULT[1].split(",")
    for name, cfg in bench.CONFIGS.items():
        assert cfg["metric"] in ("l2", "angular") and cfg["n"] > 0 and cfg["dim"] > 0
        assert cfg["ef"] > 0 or (cfg["sweep"] == sorted(cfg["sweep"]) and len(cfg["sweep"]) == len(set(cfg["sweep"])) and cfg["sweep"]), name
        assert cfg.get("dtype", "float32") in ("float32", "uint8")
    assert bench.recorded_traffic("c2", "float32", 1, 1, 1) is None  # nothing recorded for a workload nobody profiled


def test_resident_per_cu_follows_the_measured_lds_granules():
    # (LDS bytes per single-wave workgroup, what the occupancy API reports, resident workgroups per CU MEASURED on an MI355X by
    #  tools/dev/probes/lds_granule.cpp in round 4: gfx950 hands LDS out in 1280-byte granules, which the API does not know)
    measured = [(1024, 32, 32), (4096, 32, 32), (5120, 32, 32), (6400, 25, 25), (7600, 21, 21), (7680, 21, 21), (7681, 21, 18),
                (7712, 21, 18), (8192, 20, 18), (8960, 18, 18), (8961, 18, 16), (9100, 18, 16), (10144, 16, 16), (10240, 16, 16),
                (10241, 15, 14), (12768, 12, 12), (12800, 12, 12), (12801, 12, 11), (13584, 12, 11), (14080, 11, 11), (15552, 10, 9),
                (16640, 9, 9), (20480, 8, 8), (32768, 5, 4)]
    for lds, api, resident in measured:
        assert bench.resident_per_cu(api, lds) == resident, (lds, api, resident)
    assert bench.resident_per_cu(16, 7680) == 16  # registers / the grid limit first
