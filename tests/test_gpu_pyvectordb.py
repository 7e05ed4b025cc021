"""GPU: the PyVectorDB surface (src/python_bindings.rs:42-307) end to end — insert, staging file,
optimize (builds the RaBitQ index), search / search_with_stats / get_stats — with the results
checked against the CPU oracle on the index that optimize() built."""
import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.pyvectordb import PyVectorDB

from helpers import clustered, queries_like

pytestmark = pytest.mark.gpu


def test_pyvectordb_end_to_end(gpu, tmp_path, capsys):
    from oracle import oracle as orc

    x = clustered(5000, 32, 40, 11, sigma=0.2)
    path = str(tmp_path / "vdb.bin")
    db = PyVectorDB(32, branching_factor=8, target_leaf_size=40, metric="Cosine", temp_file_path=path, device_id=gpu)
    assert db.insert_embeddings(x[:3000].tolist()) == 3000
    assert db.insert_embeddings(x[3000:].tolist()) == 2000
    assert len(db) == 5000
    dt = db.optimize()
    assert dt > 0 and "Building index with 5000 vectors..." in capsys.readouterr().out
    assert len(db) == 5000
    # MmapVectorStore::create wrote the raw rows in insertion order (src/storage/mmap.rs:59-98)
    assert np.array_equal(np.fromfile(path, dtype="<f4").reshape(5000, 32), x)

    fi = db.index.flat
    oi = orc.OracleIndex(fi)
    q = queries_like(x, 16, 5, noise=0.05)
    batch = db.search_batch(q, 10, 4, 10)
    for i in range(q.shape[0]):
        t = oi.search(q[i], 10, 4, 10, trace=True)
        got = db.search(q[i].tolist(), k=10, probes=4, rerank_factor=10)
        assert got == batch[i]
        assert [g[0] for g in got] == [int(v) for v in t["ids"][:t["count"]]]
        assert np.array_equal(np.array([g[1] for g in got], np.float32).view(np.uint32), t["dist"][:t["count"]].view(np.uint32))
    assert [r[0] for r in db.search(x[17].tolist(), k=1, probes=8, rerank_factor=10)] == [17]

    res, st = db.search_with_stats(q[0].tolist(), k=10, probes=4, rerank_factor=10)
    t = oi.search(q[0], 10, 4, 10, _ffi.SEARCH_STATS_VARIANT, trace=True)
    assert sorted(st) == sorted(["total_leaves", "leaves_searched", "total_vectors", "vectors_scanned_rabitq",
                                 "vectors_reranked_full", "tree_depth"])
    assert st["leaves_searched"] == len(t["leaves"]) and st["vectors_scanned_rabitq"] == len(t["cand_pos"])
    assert st["vectors_reranked_full"] == min(100, len(t["cand_pos"])) and st["total_vectors"] == 5000
    assert [r[0] for r in res] == [int(v) for v in t["ids"][:t["count"]]]

    gs = db.get_stats()
    assert gs["num_vectors"] == 5000 and gs["dimension"] == 32 and gs["num_nodes"] == fi.n_nodes
    assert gs["num_leaves"] == st["total_leaves"] and gs["max_depth"] == st["tree_depth"] == fi.n_levels() - 1
    assert gs["disk_usage_bytes"] == 5000 * 32 * 4 and gs["max_leaf_size"] >= 1
    assert repr(db) == f"PyVectorDB(vectors=5000, dimension=32, depth={gs['max_depth']}, leaves={gs['num_leaves']})"
    with pytest.raises(ValueError, match="Query dimension 31 doesn't match index dimension 32"):
        db.search([0.0] * 31)


def test_pyvectordb_optimize_from_staging_file(gpu, tmp_path, capsys):
    """A second process that never saw the inserts builds from `<temp_file_path>.staging` (:126-136)."""
    x = clustered(1200, 16, 10, 3)
    path = str(tmp_path / "shared.bin")
    writer = PyVectorDB(16, branching_factor=6, target_leaf_size=50, temp_file_path=path, device_id=gpu)
    writer.insert_embeddings(x[:700].tolist())
    writer.insert_embeddings(x[700:].tolist())
    other = PyVectorDB(16, branching_factor=6, target_leaf_size=50, temp_file_path=path, device_id=gpu)
    assert len(other) == 0
    other.optimize()
    assert "Loaded 1200 vectors from staging file" in capsys.readouterr().out
    assert len(other) == 1200
    writer.optimize()
    q = queries_like(x, 8, 1)
    assert other.search_batch(q, 5, 3, 10) == writer.search_batch(q, 5, 3, 10)
