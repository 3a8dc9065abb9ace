"""The header-only C++ host API (include/flatnav/**) and the construct / query command-line pair, exercised WITHOUT
Python in between: a C++ test program in the shape of the reference's test_serialization.cpp, and the two tools run
on .npy files like the reference's tools/construct_npy.cpp / tools/query_npy.cpp."""
import os
import re
import subprocess

import numpy as np
import pytest

from flatnav_amd import build_host, datasets as ds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api_test_program():
    return build_host.build_program(os.path.join(ROOT, "tests", "cpp", "test_index_api.cpp"),
                                    os.path.join(ROOT, "tests", "cpp", "test_index_api.bin"))


def test_cpp_program_builds_saves_loads_without_a_gpu(api_test_program, tmp_path):
    out = subprocess.run([api_test_program, "--no-gpu", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr


def test_npy_reader_round_trip(tmp_path):
    # the tools' own .npy reader against numpy-written files (versions 1.0 headers; float32 / uint8 / int64)
    tools = build_host.build_tools()
    assert all(os.path.exists(p) for p in tools.values())
    out = subprocess.run([tools["flatnav_query"]], capture_output=True, text=True)
    assert out.returncode != 0 and "Usage" in out.stderr


@pytest.mark.gpu
def test_cpp_program_searches_on_the_gpu(api_test_program, tmp_path):
    out = subprocess.run([api_test_program, "--gpu", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True], ids=["host_builder", "device_builder"])
def test_construct_then_query_command_line_pair(tmp_path, device):
    import flatnav_amd as flatnav

    tools = build_host.build_tools()
    N, NQ, K = 20000, 500, 10
    X, Q = ds.sift_like(N, NQ)
    gt = ds.exact_topk_l2(X, Q, 100).astype(np.int32)
    np.save(tmp_path / "train.npy", X)
    np.save(tmp_path / "test.npy", Q)
    np.save(tmp_path / "gt.npy", gt)
    index_file = str(tmp_path / "index.bin")
    cmd = [tools["flatnav_construct"], "0", "0", str(tmp_path / "train.npy"), "32", "100", "4", index_file]
    out = subprocess.run(cmd + (["--device"] if device else []), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    assert os.path.getsize(index_file) == 60 + N * (128 * 4 + 32 * 4 + 4)
    out = subprocess.run([tools["flatnav_query"], "0", index_file, str(tmp_path / "test.npy"), str(tmp_path / "gt.npy"),
                          "50,100,200", str(K), "0", "0"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    recalls = [float(m) for m in re.findall(r"Mean Recall: ([0-9.]+)", out.stdout)]
    assert len(recalls) == 3 and recalls[0] > 0.9 and recalls[1] >= 0.98 and recalls[2] >= recalls[1] - 1e-9
    # the same file through the Python surface answers identically to the tool's numbers
    index = flatnav.index.IndexL2Float.load_index(index_file)
    _, labels = index.search(Q, K, 100)
    assert abs(ds.recall_at_k(labels, gt[:, :K]) - recalls[1]) < 1e-6
    # per-query protocol of the reference (query_npy.cpp:41-66): same recall, one call per query
    out = subprocess.run([tools["flatnav_query"], "0", index_file, str(tmp_path / "test.npy"), str(tmp_path / "gt.npy"),
                          "100", str(K), "0", "0", "--per-query"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and abs(float(re.findall(r"Mean Recall: ([0-9.]+)", out.stdout)[0]) - recalls[1]) < 1e-9
