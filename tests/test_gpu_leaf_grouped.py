"""GPU: the Hamming scan leaf by leaf (leaf_hamming_kernel + the per-query selection from handed-over distances,
chosen by the library for wide codes and batches that reach a leaf several times; CUDDLE_LEAF_GROUPED forces the
choice) must give what the per-query scan gives, bit for bit, and what the oracle gives
(reference: collect_leaf_candidates / compute_hamming_distances / select_top_candidates,
src/index/hierarchical.rs:591-655)."""
import numpy as np
import pytest
import torch

from vectordb_amd.index import DeviceIndex
from vectordb_amd.sharded import exact_search_in_process

from helpers import clustered, queries_like, small_index

pytestmark = pytest.mark.gpu


def _same(a, b):
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)


@pytest.mark.parametrize("d,metric", [(40, "l2"), (128, "l2"), (130, "cosine"), (256, "dot"), (512, "l2"), (960, "l2"), (1000, "l2")])
def test_leaf_by_leaf_equals_per_query(gpu, monkeypatch, d, metric):
    # widths with and without an instantiation of the leaf kernel (1000-d = 16 words has one, 130-d = 3 words too)
    from oracle import oracle as orc
    x = clustered(5000, d, 40, 3, sigma=0.25)
    from vectordb_amd.builder import build_index
    fi = build_index(x, 8, 40, metric, 6, seed=2)
    q = queries_like(x, 400, 7, noise=0.1)   # 400 queries over ~125 leaves: many leaves get more than 64 queries
    dev = DeviceIndex(fi, gpu)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CUDDLE_LEAF_GROUPED", mode)
        res[mode] = [dev.search_batch(q, 10, 6, rf) for rf in (1, 7, 1000)]
        res[mode].append(dev.search_batch(q, 10, 6, 5, want_stats=True)[3])
    monkeypatch.delenv("CUDDLE_LEAF_GROUPED")
    for a, b in zip(res["0"][:3], res["1"][:3]):
        _same(a, b)
    assert res["0"][3] == res["1"][3]
    oi = orc.OracleIndex(fi)
    rc, ids, dd, cnt = oi.search_batch(q[:60], 10, 6, 7)
    assert rc == 0
    _same((ids, dd, cnt), tuple(a[:60] for a in res["1"][1]))


def test_leaf_by_leaf_sharded_and_spilled(gpu, monkeypatch):
    """Leaf shards (empty leaves in every rank's lists) through the histogram steps, and queries with more candidates
    than the scan keeps in LDS (the distances stay in the spill rows)."""
    monkeypatch.setenv("CUDDLE_LEAF_GROUPED", "1")
    x, fi = small_index(n=6000, d=320, n_clusters=40, branching=8, leaf=40)
    q = queries_like(x, 96, 3)
    full = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(q).cuda(gpu)
    ref = full.search_batch_device(dq, 10, 6, 8)
    shards = [DeviceIndex(fi.shard(r, 3), gpu) for r in range(3)]
    got = exact_search_in_process(shards, dq, 10, 6, 8)
    torch.cuda.synchronize()
    _same([t.cpu().numpy() for t in ref], [t.cpu().numpy() for t in got])
    # 30 000 candidates per query (> 24 576 kept in LDS): 60 leaves of 500 vectors, all probed
    rng = np.random.default_rng(1)
    n, d, nl = 30000, 512, 60
    xs = rng.standard_normal((n, d)).astype(np.float32)
    from vectordb_amd import _ffi
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    big = FlatIndex(dim=d, metric=_ffi.METRIC_L2, code_kind=_ffi.CODE_BINARY, centroids=xs.reshape(nl, n // nl, d).mean(axis=1),
                    child_offsets=np.zeros(nl + 1, np.uint32), children=np.zeros(0, np.uint32), roots=np.arange(nl, dtype=np.uint32),
                    vec_offsets=(np.arange(nl + 1) * (n // nl)).astype(np.uint64), ids=np.arange(n, dtype=np.uint64), vectors=xs,
                    threshold=0.0, codes=pack_binary_codes(xs, 0.0))
    dev = DeviceIndex(big, gpu)
    qs = xs[:80] + 0.1 * rng.standard_normal((80, d)).astype(np.float32)
    a = dev.search_batch(qs, 10, 60, 30)
    monkeypatch.setenv("CUDDLE_LEAF_GROUPED", "0")
    b = dev.search_batch(qs, 10, 60, 30)
    _same(a, b)
