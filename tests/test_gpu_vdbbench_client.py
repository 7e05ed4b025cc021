"""GPU: the harness client end to end (docs/vectordbbench_integration/VECTORDBBENCH_INTEGRATION.md:59-108) — harness ids
in, harness ids out, the same answers as PyVectorDB gives for array indices, and a second process' view (unpickled:
configuration only) that rebuilds from the staging file and the id mapping and answers the same."""
import pickle

import numpy as np
import pytest

from vectordb_amd.vdbbench_client import CuddleCaseConfig, CuddleVectorDB

from helpers import clustered, queries_like

pytestmark = pytest.mark.gpu


def test_client_end_to_end(gpu, tmp_path):
    x = clustered(4000, 48, 30, 5, sigma=0.2)
    ids = np.random.default_rng(2).permutation(1_000_000)[:4000].astype(np.int64)
    cfg = CuddleCaseConfig(branching_factor=8, target_leaf_size=40, probes=8, rerank_factor=10, metric="Cosine")
    db = CuddleVectorDB(48, {"path": str(tmp_path / "bench.bin"), "device_id": gpu}, cfg, drop_old=True)
    with db.init():
        for i in range(0, 4000, 1000):
            assert db.insert_embeddings(x[i:i + 1000].tolist(), ids[i:i + 1000].tolist()) == (1000, None)
        db.optimize()
        q = queries_like(x, 32, 9, noise=0.05)
        xn = x / np.linalg.norm(x, axis=1, keepdims=True)
        hits = 0
        first = []
        for i in range(q.shape[0]):
            got = db.search_embedding(q[i].tolist(), k=10)
            raw = db.db.search(q[i].tolist(), 10, 8, 10)
            assert got == [int(ids[r[0]]) for r in raw]
            truth = np.argsort(-(xn @ (q[i] / np.linalg.norm(q[i]))), kind="stable")[:10]
            hits += len(set(got) & set(ids[truth].tolist()))
            first.append(got)
        assert hits >= 0.9 * 10 * q.shape[0]
        assert db.search_embeddings(q, k=10) == first
        assert db.search_embedding(x[123].tolist(), k=1) == [int(ids[123])]
    other = pickle.loads(pickle.dumps(db))
    with other.init():
        assert other.search_embeddings(q, k=10) == first
