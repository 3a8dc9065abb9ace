"""flatnav/util/StlExact.h (shared by the HIP kernel and the host builder) must perform exactly
libstdc++'s heap / introsort element moves -- checked against the real std::priority_queue and
std::sort under heavy ties by a small C++ program (tests/stl_exact_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stl_exact_matches_libstdcxx(tmp_path):
    exe = str(tmp_path / "stl_exact_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "stl_exact_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "stl_exact: OK" in out.stdout
