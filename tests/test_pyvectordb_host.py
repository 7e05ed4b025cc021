"""The host-side behaviour of PyVectorDB that needs no GPU: argument checks, error messages,
counters and the staging file (src/python_bindings.rs:52-113, 278-430)."""
import struct

import numpy as np
import pytest

from vectordb_amd.pyvectordb import PyVectorDB


def test_constructor_metrics(tmp_path):
    for m in ("L2", "Cosine", "InnerProduct", "DotProduct"):
        PyVectorDB(8, metric=m, temp_file_path=str(tmp_path / "x.bin"))
    with pytest.raises(ValueError, match="Unknown metric 'l2'. Use 'L2', 'Cosine', or 'InnerProduct'"):
        PyVectorDB(8, metric="l2")


def test_insert_validates_and_stages(tmp_path):
    path = str(tmp_path / "db.bin")
    db = PyVectorDB(4, temp_file_path=path)
    assert len(db) == 0 and repr(db) == "PyVectorDB(dimension=4, not_built, accumulated_vectors=0)"
    assert db.insert_embeddings([[1, 2, 3, 4], [5, 6, 7, 8]]) == 2
    with pytest.raises(ValueError, match="Vector 1 has dimension 3, expected 4"):
        db.insert_embeddings([[1, 2, 3, 4], [1, 2, 3]])
    assert db.insert_embeddings([[9, 9, 9, 9]]) == 1
    assert len(db) == 3 and "accumulated_vectors=3" in repr(db)
    want = np.array([[1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 9, 9]], dtype="<f4")
    assert open(path + ".staging", "rb").read() == struct.pack("<QQ", 3, 4) + want.tobytes()


def test_calls_before_optimize(tmp_path):
    db = PyVectorDB(4, temp_file_path=str(tmp_path / "db.bin"))
    with pytest.raises(RuntimeError, match=r"Index not built. Call optimize\(\) first"):
        db.search([0, 0, 0, 0])
    with pytest.raises(ValueError, match="Query dimension 3 doesn't match index dimension 4"):
        db.search([0, 0, 0])
    with pytest.raises(RuntimeError, match="Index not built$"):
        db.get_stats()
    with pytest.raises(ValueError, match="No vectors to build index from"):
        db.optimize()
