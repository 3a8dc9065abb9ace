"""C-ABI surface checks that need no GPU: the shared library builds for gfx950, loads, and exports
exactly the symbols include/flatnav_hip.h declares.  No compute entry point is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from flatnav_amd import build

    return build.build()  # hipcc cross-compiles for gfx950 without a GPU


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "flatnav_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fnv_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from flatnav_amd import hip

    assert _declared_symbols() == sorted(hip.C_ABI_SYMBOLS)


def test_every_option_the_library_accepts_is_documented_in_the_header():
    # fnv_set_option's names (csrc/beam_search.hip) against the option list of include/flatnav_hip.h
    src = open(os.path.join(ROOT, "flatnav_amd", "csrc", "beam_search.hip")).read()
    body = src[src.index("int fnv_set_option("):]
    body = body[:body.index("\nint ", 10)]
    names = sorted(set(re.findall(r'n == "([a-z_0-9]+)"', body)))
    assert len(names) >= 25, names
    hdr = open(os.path.join(ROOT, "include", "flatnav_hip.h")).read()
    assert [n for n in names if '"%s"' % n not in hdr] == []


def test_library_exports_every_declared_symbol(libpath):
    lib = C.CDLL(libpath)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.fnv_version.restype = C.c_char_p
    assert b"gfx950" in lib.fnv_version()
    lib.fnv_last_error.restype = C.c_char_p
    assert lib.fnv_last_error() is not None


def test_library_exports_nothing_else(libpath):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == _declared_symbols()  # kernels and helpers stay hidden (-fvisibility=hidden)


def test_library_contains_gfx950_code_object(libpath):
    blob = open(libpath, "rb").read()
    assert b"gfx950" in blob and b"beam_search_kernel" in blob


def test_argument_validation_needs_no_device(libpath):
    from flatnav_amd import hip

    L = hip.lib()
    h = C.c_void_p()
    # unsupported data type -> runtime error (reference: "Unsupported data type", bindings.cpp:498-499)
    assert L.fnv_index_alloc(8, 100, 3, 0, 16, 0, C.byref(h)) == hip.FNV_ERR_RUNTIME
    # empty index / bad metric -> invalid argument
    assert L.fnv_index_alloc(8, 0, 9, 0, 16, 0, C.byref(h)) == hip.FNV_ERR_INVALID
    assert L.fnv_index_alloc(8, 100, 9, 7, 16, 0, C.byref(h)) == hip.FNV_ERR_INVALID
    assert L.fnv_search_batch(None, None, 1, 1, 1, 1, None, None, None, None, None) == hip.FNV_ERR_INVALID
    assert b"index is null" in L.fnv_last_error()


def test_row_layout_rules_need_no_device(libpath, monkeypatch):
    # fnv_row_layout: the stride / split-row rule of the vector table (csrc/beam_search.hip row_layout), a pure function of
    # (dim, element type, capacity) and three environment variables
    from flatnav_amd import hip

    for k in ("FLATNAV_ROW_PAD_PCT", "FLATNAV_SPLIT_ROWS", "FLATNAV_SPLIT_TAIL_MAX_MB"):
        monkeypatch.delenv(k, raising=False)
    assert hip.row_layout(128, "float32", 10**6) == (512, 0)      # whole lines already
    assert hip.row_layout(128, "uint8", 10**6) == (128, 0)
    assert hip.row_layout(768, "float32", 10**7) == (3072, 0)
    assert hip.row_layout(100, "float32", 1_183_514) == (384, 16)  # GloVe shape: three lines + a 16-byte tail in the side table
    assert hip.row_layout(104, "float32", 1000) == (384, 32) and hip.row_layout(97, "float32", 1000) == (384, 16)
    assert hip.row_layout(400, "uint8", 1000) == (384, 16) and hip.row_layout(410, "int8", 1000) == (384, 32)
    assert hip.row_layout(96, "float32", 1000) == (384, 0)         # exactly three lines: nothing to split
    assert hip.row_layout(105, "float32", 1000) == (512, 0)        # 48 bytes over: padded (<= 30 %), not split
    assert hip.row_layout(200, "float32", 1000) == (896, 0)        # six lines + 32 bytes: the kernels split three-line rows only
    assert hip.row_layout(37, "int8", 1000) == (48, 0)             # padding to a line would cost 167 %: 16-byte stride
    assert hip.row_layout(100, "float32", 5_000_000) == (512, 0)   # an 80 MB side table would not stay cached: padded
    monkeypatch.setenv("FLATNAV_SPLIT_TAIL_MAX_MB", "128")
    assert hip.row_layout(100, "float32", 5_000_000) == (384, 16)
    monkeypatch.setenv("FLATNAV_SPLIT_ROWS", "0")
    assert hip.row_layout(100, "float32", 1000) == (512, 0)
    monkeypatch.setenv("FLATNAV_ROW_PAD_PCT", "0")
    assert hip.row_layout(100, "float32", 1000) == (400, 0)
    with pytest.raises(RuntimeError):  # "Unsupported data type" (reference bindings.cpp:498-499): uint64 is not an index element type
        hip.check(hip.lib().fnv_row_layout(100, 3, 10, C.byref(C.c_uint32()), C.byref(C.c_uint32())))
    with pytest.raises(ValueError):
        hip.row_layout(100, "float32", 0)


def test_search_without_gpu_fails_loudly():
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; this checks the no-device error path")
    import flatnav_amd as flatnav

    ix = flatnav.index.create("l2", 8, 64, 4)
    ix.add(np.random.default_rng(0).random((64, 8), dtype=np.float32), 16)
    with pytest.raises(RuntimeError):
        ix.search(np.zeros((2, 8), dtype=np.float32), 3, 10)  # never falls back to a CPU search
