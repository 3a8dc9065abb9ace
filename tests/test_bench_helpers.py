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
