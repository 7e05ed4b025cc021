"""The harness client's host logic with the GPU index replaced by a stand-in: id mapping, its file, pickling that carries
the configuration only, errors returned (not raised) from insert_embeddings
(docs/vectordbbench_integration/VECTORDBBENCH_INTEGRATION.md:100-108, :356-390)."""
import pickle

import numpy as np
import pytest

from vectordb_amd import vdbbench_client as vc
from vectordb_amd.datasets import read_staging
from vectordb_amd.pyvectordb import PyVectorDB as _RealDB


class _BruteIndex:
    def __init__(self, x):
        self.x = x

    def search_batch(self, q, k, probes, rf):
        d = ((q[:, None, :] - self.x[None]) ** 2).sum(-1)
        ids = np.argsort(d, axis=1, kind="stable")[:, :k].astype(np.uint64)
        return ids, np.take_along_axis(d, ids.astype(np.int64), 1).astype(np.float32), np.full(q.shape[0], ids.shape[1], np.int32)


class _StandInDB(vc.PyVectorDB):
    """PyVectorDB with optimize() building an exhaustive CPU index: the inserts, the staging file and every check stay
    the real ones."""

    def optimize(self):
        if self._pending_rows:
            x = np.concatenate(self._pending)
            self._pending, self._pending_rows = [], 0
        else:
            x = read_staging(self._staging_path, self.dimension)
        self.index = _BruteIndex(x)
        self._n = x.shape[0]
        return 0.0

    def __len__(self):
        return self._n if self.index is not None else self._pending_rows

    def search(self, query, k=10, probes=100, rerank_factor=10):
        ids, dist, cnt = self.index.search_batch(np.asarray(query, np.float32)[None], k, probes, rerank_factor)
        return [(int(i), float(d)) for i, d in zip(ids[0], dist[0])]


@pytest.fixture
def client(tmp_path, monkeypatch):
    monkeypatch.setattr(vc, "PyVectorDB", _StandInDB)
    return vc.CuddleVectorDB(4, {"path": str(tmp_path / "bench.bin")}, vc.CuddleCaseConfig(8, 16, 4, 5, "L2"), drop_old=True)


def test_ids_are_the_harness_ids_and_survive_pickling(client, tmp_path):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 4)).astype(np.float32)
    ids = rng.permutation(100000)[:50] + 7
    with client.init():
        assert client.insert_embeddings(x[:30].tolist(), ids[:30].tolist()) == (30, None)
        assert client.insert_embeddings(x[30:], ids[30:]) == (20, None)
        client.optimize()
        got = client.search_embedding(x[12].tolist(), k=3)
        assert got[0] == ids[12] and len(got) == 3 and all(isinstance(v, int) for v in got)
        batch = client.search_embeddings(x[:5], k=2)
        assert [b[0] for b in batch] == [int(v) for v in ids[:5]]
    assert np.array_equal(np.load(str(tmp_path / "bench.bin.idmap.npy")), ids)
    blob = pickle.dumps(client)
    assert len(blob) < 600  # the configuration, not the vectors
    other = pickle.loads(blob)  # a process that did not see the inserts
    assert other.db is None
    with pytest.raises(RuntimeError, match="Index not optimized"):
        other.search_embedding(x[0].tolist(), 1)
    with other.init():
        assert other.search_embedding(x[41].tolist(), k=1) == [int(ids[41])]
        assert other.case_config == client.case_config


def test_insert_errors_are_returned(client):
    with client.init():
        n, err = client.insert_embeddings([[1, 2, 3, 4], [1, 2, 3, 4]], [5])
        assert n == 0 and isinstance(err, ValueError)
        n, err = client.insert_embeddings([[1, 2, 3]], [5])
        assert n == 0 and isinstance(err, ValueError) and "dimension 3, expected 4" in str(err)
        with pytest.raises(ValueError, match="No vectors to build index from"):
            _RealDB.optimize(client.db)  # nothing inserted, no staging file


def test_drop_old_removes_the_files(tmp_path, monkeypatch):
    monkeypatch.setattr(vc, "PyVectorDB", _StandInDB)
    cfg = {"path": str(tmp_path / "b.bin")}
    a = vc.CuddleVectorDB(4, cfg)
    with a.init():
        a.insert_embeddings(np.eye(4, dtype=np.float32), [10, 11, 12, 13])
        a.optimize()
    assert (tmp_path / "b.bin.staging").exists() and (tmp_path / "b.bin.idmap.npy").exists()
    b = vc.CuddleVectorDB(4, cfg, drop_old=True)
    assert not (tmp_path / "b.bin.staging").exists() and not (tmp_path / "b.bin.idmap.npy").exists()
    with b.init():
        assert b.db.index is None
    assert not b.need_normalize_cosine()
