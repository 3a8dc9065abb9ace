"""Dataset readers (flatnav_amd.io) against files written with numpy in the documented layouts."""
import numpy as np
import pytest

from flatnav_amd import io as fio


def _write_vecs(path, a):
    n, d = a.shape
    rec = np.empty((n, 4 + d * a.dtype.itemsize), dtype=np.uint8)
    rec[:, :4] = np.array([d], dtype=np.int32).view(np.uint8)
    rec[:, 4:] = np.ascontiguousarray(a).view(np.uint8).reshape(n, -1)
    rec.tofile(path)


@pytest.mark.parametrize("ext,dt", [(".fvecs", np.float32), (".ivecs", np.int32), (".bvecs", np.uint8)])
def test_vecs_roundtrip(tmp_path, ext, dt):
    rng = np.random.default_rng(0)
    a = (rng.random((37, 19)) * 100).astype(dt)
    p = str(tmp_path / ("x" + ext))
    _write_vecs(p, a)
    assert np.array_equal(fio.read_vecs(p), a)
    assert np.array_equal(fio.load_matrix(p, rows=(5, 12)), a[5:12])
    assert fio.load_matrix(p, rows=(30, 100)).shape == (7, 19)


@pytest.mark.parametrize("ext,dt", [(".fbin", np.float32), (".u8bin", np.uint8), (".i8bin", np.int8)])
def test_bin_roundtrip(tmp_path, ext, dt):
    rng = np.random.default_rng(1)
    a = (rng.random((50, 24)) * 100 - 50).astype(dt)
    p = str(tmp_path / ("x" + ext))
    with open(p, "wb") as f:
        np.array(a.shape, dtype=np.uint32).tofile(f)
        a.tofile(f)
    assert np.array_equal(fio.read_bin(p), a)
    assert np.array_equal(fio.load_matrix(p, rows=(10, 20)), a[10:20])


def test_npy_ground_truth_and_errors(tmp_path):
    rng = np.random.default_rng(2)
    X, Q = rng.random((20, 8), dtype=np.float32) + 0.1, rng.random((5, 8), dtype=np.float32) + 0.1
    ids = rng.integers(0, 20, (5, 10)).astype(np.uint32)
    dist = rng.random((5, 10), dtype=np.float32)
    np.save(tmp_path / "train.npy", X)
    np.save(tmp_path / "q.npy", Q)
    with open(tmp_path / "gt.bin", "wb") as f:
        np.array(ids.shape, dtype=np.uint32).tofile(f)
        ids.tofile(f)
        dist.tofile(f)
    Xl, Ql, G = fio.load_dataset(str(tmp_path / "train.npy"), str(tmp_path / "q.npy"), str(tmp_path / "gt.bin"),
                                 normalize=True)
    assert np.allclose(np.linalg.norm(Xl, axis=1), 1, atol=1e-6) and np.allclose(np.linalg.norm(Ql, axis=1), 1, atol=1e-6)
    assert G.dtype == np.int32 and np.array_equal(G, ids.astype(np.int32))
    gi, gd = fio.read_ground_truth_bin(str(tmp_path / "gt.bin"))
    assert np.array_equal(gd, dist)
    with pytest.raises(FileNotFoundError):
        fio.load_matrix(str(tmp_path / "missing.npy"))
    with pytest.raises(ValueError):
        fio.load_matrix(str(tmp_path / "train.npy"), rows=(5, 2))
    (tmp_path / "x.txt").write_text("1")
    with pytest.raises(ValueError):
        fio.load_matrix(str(tmp_path / "x.txt"))
