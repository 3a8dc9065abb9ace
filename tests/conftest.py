import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Lines the tests want in the run's terminal summary -- it is printed under -q as well, so the driver's log tail says which
# sizes ran ("10M x 768 / 50M x 128 FULL SIZE") and which id-equality fractions were measured.
SUMMARY_LINES = []


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if SUMMARY_LINES:
        terminalreporter.write_sep("-", "flatnav_amd: sizes and measured parity fractions")
        for line in SUMMARY_LINES:
            terminalreporter.write_line(line)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


# Result-pinning tests first, anything that measures time or drives many host threads last: under `-x` a late
# failure can then never hide the parity suites.  Files not named keep their alphabetical place in the middle.
_ORDER = ["test_golden", "test_gpu_parity", "test_gpu_round5", "test_gpu_round6", "test_gpu_configs", "test_cpp_api",
          "test_gpu_python_api", "test_gpu_round4", "test_gpu_round3", "test_gpu_device_build", "test_gpu_fullsize",
          "test_gpu_bench", "test_gpu_multi_device"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _ORDER:
            return _ORDER.index(name) + (0 if _ORDER.index(name) < 5 else 1000)
        return 500

    items.sort(key=rank)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def ref(oracle_mod):
    R = oracle_mod.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref was never built (needs /root/reference in the dev container)")
    return R
