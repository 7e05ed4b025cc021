"""Readers and writers of the on-disk formats either side of the search path: .fvecs / .ivecs
(datasets/sift/loader.rs:10-87), the generator's .bin (datasets/generate_datasets.rs:165-189), the raw
vector file (src/storage/mmap.rs:26-98) and the Python binding's staging file
(src/python_bindings.rs:313-430). Files are written here byte by byte with struct, read with the
package's readers, and the package's writers must produce the same bytes."""
import os
import struct

import numpy as np
import pytest

from vectordb_amd import datasets as ds


def _rows(n, d, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return rng.standard_normal((n, d)).astype(np.float32)
    return rng.integers(0, 1 << 20, size=(n, d)).astype(np.int32)


def test_fvecs_bytes_and_round_trip(tmp_path):
    x = _rows(7, 5, 1)
    raw = b"".join(struct.pack("<i", 5) + struct.pack("<5f", *row) for row in x)
    p = tmp_path / "a.fvecs"
    p.write_bytes(raw)
    got = ds.read_fvecs(str(p))
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), x.view(np.uint32))
    assert np.array_equal(ds.read_fvecs(str(p), limit=3), x[:3])  # loader.rs:24-27 stops at the limit
    ds.write_fvecs(str(tmp_path / "b.fvecs"), x)
    assert (tmp_path / "b.fvecs").read_bytes() == raw


def test_ivecs_bytes_and_round_trip(tmp_path):
    g = _rows(6, 100, 2, np.int32)
    raw = b"".join(struct.pack("<i", 100) + struct.pack("<100i", *row) for row in g)
    p = tmp_path / "gt.ivecs"
    p.write_bytes(raw)
    got = ds.read_ivecs(str(p))
    assert got.dtype == np.int32 and np.array_equal(got, g)
    assert np.array_equal(ds.read_ivecs(str(p), limit=2), g[:2])
    ds.write_ivecs(str(tmp_path / "w.ivecs"), g)
    assert (tmp_path / "w.ivecs").read_bytes() == raw


def test_vecs_empty_and_inconsistent_headers(tmp_path):
    e = tmp_path / "empty.fvecs"
    e.write_bytes(b"")
    assert ds.read_fvecs(str(e)).shape == (0, 0) and ds.read_ivecs(str(e)).shape == (0, 0)
    bad = tmp_path / "bad.fvecs"
    bad.write_bytes(struct.pack("<i3f", 3, 1, 2, 3) + struct.pack("<i3f", 2, 1, 2, 3))
    with pytest.raises(ValueError):
        ds.read_fvecs(str(bad))
    with pytest.raises(ValueError):
        ds.read_ivecs(str(bad))


def test_generator_bin_bytes_and_round_trip(tmp_path):
    x = _rows(9, 4, 3)
    raw = struct.pack("<II", 9, 4) + x.astype("<f4").tobytes()
    p = tmp_path / "g.bin"
    p.write_bytes(raw)
    assert np.array_equal(ds.read_generator_bin(str(p)), x)
    ds.write_generator_bin(str(tmp_path / "w.bin"), x)
    assert (tmp_path / "w.bin").read_bytes() == raw


def test_raw_vector_file_size_check(tmp_path):
    x = _rows(11, 8, 4)
    p = tmp_path / "v.bin"
    x.astype("<f4").tofile(str(p))
    got = ds.read_raw_vectors(str(p), 8, 11)
    assert np.array_equal(np.asarray(got), x)
    with pytest.raises(ValueError, match="File size mismatch"):  # mmap.rs:35-42
        ds.read_raw_vectors(str(p), 8, 12)


def test_staging_file_appends_and_rewrites_count(tmp_path):
    p = str(tmp_path / "db.bin.staging")
    assert ds.read_staging(p, 6).shape == (0, 6)  # python_bindings.rs:392-394: no file, no vectors
    a, b = _rows(5, 6, 5), _rows(3, 6, 6)
    assert ds.append_staging(p, a, 6) == 5
    assert open(p, "rb").read() == struct.pack("<QQ", 5, 6) + a.tobytes()
    assert ds.append_staging(p, b, 6) == 8
    assert open(p, "rb").read() == struct.pack("<QQ", 8, 6) + a.tobytes() + b.tobytes()
    assert np.array_equal(ds.read_staging(p, 6), np.concatenate([a, b]))
    assert ds.append_staging(p, np.zeros((0, 6), np.float32), 6) == 8
    with pytest.raises(ValueError, match="Dimension mismatch: expected 7, got 6"):  # :410-414
        ds.read_staging(p, 7)
    with pytest.raises(ValueError, match="Dimension mismatch"):
        ds.append_staging(p, _rows(1, 7, 7), 7)
    with open(p, "r+b") as f:  # a header that promises more than the file holds: read_exact fails (:422-423)
        f.write(struct.pack("<Q", 9))
    with pytest.raises(ValueError, match="Failed to read vector data"):
        ds.read_staging(p, 6)


def test_recall_at_k_matches_definition():
    found = np.array([[1, 2, 3, 4], [9, 8, 7, 6]])
    gt = np.array([[4, 3, 0, 0], [1, 2, 3, 4]])
    assert ds.recall_at_k(found, gt, 2) == 0.0 and ds.recall_at_k(found, gt, 4) == pytest.approx(2 / 8)
