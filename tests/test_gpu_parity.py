"""GPU parity: libcuddle_hip.so (through the C ABI) against the CPU oracle on the same inputs.

Bar: Hamming distances, candidate sets, leaf lists, selected sets and ids bit-exact; fp32
distances compared exactly as well (the kernels reproduce the oracle's accumulation order), with
the north-star tolerance of 1e-5 relative as the documented fallback bound.
"""
import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.index import DeviceIndex

from helpers import clustered, queries_like, small_index

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


def _compare_full(fi, dev, q, k, probes, rf, flags=0):
    orc = _oracle()
    oi = orc.OracleIndex(fi)
    ids, dist, cnt = dev.search_batch(q, k, probes, rf, flags)
    for i in range(q.shape[0]):
        r = oi.search(q[i], k, probes, rf, flags)
        assert r["rc"] == 0
        assert cnt[i] == r["count"], (i, cnt[i], r["count"])
        n = r["count"]
        assert np.array_equal(ids[i, :n], r["ids"][:n]), (i, ids[i, :n], r["ids"][:n])
        assert np.array_equal(dist[i, :n].view(np.uint32), r["dist"][:n].view(np.uint32)), (i, dist[i, :n], r["dist"][:n])
        assert np.all(ids[i, n:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isinf(dist[i, n:]))


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_search_matches_oracle_small(gpu, metric):
    x, fi = small_index(metric=metric)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 64, 7)
    _compare_full(fi, dev, q, 10, 5, 10)
    _compare_full(fi, dev, q, 10, 3, 3)
    _compare_full(fi, dev, q, 1, 1, 1)
    _compare_full(fi, dev, q, 100, 50, 100)
    _compare_full(fi, dev, q, 10, 5, 10, _ffi.SEARCH_STATS_VARIANT)


def test_stage_outputs_match_oracle(gpu):
    orc = _oracle()
    x, fi = small_index(n=8000, d=128, n_clusters=80, branching=12, leaf=40, integer=True)
    dev = DeviceIndex(fi, gpu)
    oi = orc.OracleIndex(fi)
    q = queries_like(x, 32, 3, noise=3.0)
    probes, k, rf = 6, 10, 5
    leaves, lcnt = dev.route_batch(q, probes)
    traces = [oi.search(q[i], k, probes, rf, trace=True) for i in range(q.shape[0])]
    for i, t in enumerate(traces):
        assert lcnt[i] == len(t["leaves"])
        assert np.array_equal(leaves[i, :lcnt[i]], t["leaves"])
    pos, key, ccnt = dev.scan_batch(q, leaves, lcnt)
    for i, t in enumerate(traces):
        assert ccnt[i] == len(t["cand_pos"])
        assert np.array_equal(pos[i, :ccnt[i]], t["cand_pos"])
        assert np.array_equal(key[i, :ccnt[i]], t["cand_key"])
    ids, dist, cnt, spos, skey, sdist, scnt = dev.search_leaves_batch(q, leaves, lcnt, k, rf)
    for i, t in enumerate(traces):
        assert scnt[i] == len(t["sel_pos"])
        assert np.array_equal(spos[i, :scnt[i]], t["sel_pos"])
        assert np.array_equal(skey[i, :scnt[i]], t["sel_key"])
        assert np.array_equal(sdist[i, :scnt[i]].view(np.uint32), t["sel_dist"].view(np.uint32))
        assert np.array_equal(ids[i, :cnt[i]], t["ids"][:t["count"]])


@pytest.mark.parametrize("metric,d,shift,scale", [("l2", 64, 0.0, 1.0), ("cosine", 64, 0.0, 1.0), ("dot", 64, 0.0, 1.0),
                                                  ("l2", 37, 0.0, 1.0), ("cosine", 130, 0.0, 1.0), ("dot", 200, 0.0, 1.0),
                                                  ("l2", 96, 50.0, 1.0), ("l2", 128, 0.0, 1000.0), ("cosine", 72, 3.0, 0.01)])
def test_mfma_routing_is_exact(gpu, metric, d, shift, scale):
    """A level of > 512 nodes shared by all queries goes through the matrix-core contraction (split
    bf16 operands) + exact refinement; leaf lists must equal both the all-exact kernel's and the
    oracle's, and the approximate distances must stay within the bound the proof of exactness uses
    (also for data far from the origin and of large or small magnitude)."""
    orc = _oracle()
    x = ((clustered(60000, d, 300, 11, sigma=0.3) + np.float32(shift)) * np.float32(scale)).astype(np.float32)
    from vectordb_amd.builder import build_index
    fi = build_index(x, 100, 30, metric, 8, seed=3)
    widths = []
    cur = list(fi.roots)
    while cur:
        widths.append(len(cur))
        cur = [c for v in cur for c in fi.children[fi.child_offsets[v]:fi.child_offsets[v + 1]]]
    assert widths[0] <= 100 and widths[1] >= 512, widths
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 200, 5, noise=0.1)
    probes = 100
    lv_mfma, lc_mfma = dev.route_batch(q, probes)
    lv_exact, lc_exact = dev.route_batch(q, probes, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    assert np.array_equal(lc_mfma, lc_exact)
    for i in range(q.shape[0]):
        assert np.array_equal(lv_mfma[i, :lc_mfma[i]], lv_exact[i, :lc_exact[i]]), i
    oi = orc.OracleIndex(fi)
    for i in range(0, q.shape[0], 7):
        t = oi.search(q[i], 10, probes, 5, trace=True)
        assert np.array_equal(lv_mfma[i, :lc_mfma[i]], t["leaves"]), i
    ids, dist, cnt, st = dev.search_batch(q, 10, probes, 5, want_stats=True)
    assert st["route_fallbacks"] <= q.shape[0] // 10, st
    ratio, err = dev.route_filter_error(q, probes, 1)
    assert 0.0 < ratio < 0.5, (ratio, err)
    _compare_full(fi, dev, q[:40], 10, probes, 5)


@pytest.mark.parametrize("metric,d", [("l2", 64), ("cosine", 48), ("dot", 37), ("l2", 200), ("cosine", 272), ("dot", 300),
                                      ("cosine", 128), ("dot", 96), ("l2", 256)])
def test_grouped_routing_is_exact(gpu, metric, d):
    """Three levels: the roots are cut to `probes`, so the second level's frontier differs per query
    and (being wider than 100 nodes) is evaluated grouped by parent; leaf lists must equal the
    all-exact kernels' and the oracle's, for search() and for the always-sorting stats variant.
    probes = 40 also exercises a mixed batch: queries whose frontier happens to stay small."""
    orc = _oracle()
    x = clustered(40000, d, 200, 13, sigma=0.3)
    from vectordb_amd.builder import build_index
    fi = build_index(x, 30, 20, metric, 6, seed=5)
    assert fi.n_levels() >= 3
    dev = DeviceIndex(fi, gpu)
    oi = orc.OracleIndex(fi)
    q = queries_like(x, 150, 9, noise=0.1)
    for probes in (8, 40):
        for flags in (0, _ffi.SEARCH_STATS_VARIANT):
            lv_g, lc_g = dev.route_batch(q, probes, flags | _ffi.SEARCH_ROUTE_GROUPED)
            lv_e, lc_e = dev.route_batch(q, probes, flags | _ffi.SEARCH_ROUTE_EXACT_ONLY)
            assert np.array_equal(lc_g, lc_e)
            for i in range(q.shape[0]):
                assert np.array_equal(lv_g[i, :lc_g[i]], lv_e[i, :lc_e[i]]), (probes, flags, i)
            for i in range(0, q.shape[0], 11):
                t = oi.search(q[i], 10, probes, 5, flags, trace=True)
                assert np.array_equal(lv_g[i, :lc_g[i]], t["leaves"]), (probes, flags, i)
    _compare_full(fi, dev, q[:30], 10, 8, 5, _ffi.SEARCH_ROUTE_GROUPED)
    _compare_full(fi, dev, q[:30], 10, 8, 5, _ffi.SEARCH_ROUTE_GROUPED | _ffi.SEARCH_STATS_VARIANT)


def test_grouped_routing_proof_failures_take_the_exact_path(gpu, monkeypatch):
    """With no spare candidates (CUDDLE_ROUTE_MARGIN=0) the exactness proof of the grouped level fails for
    most queries. Their frontiers exist only as segment tables at that point and must be written out for
    the exact kernels; the leaf lists must still be the all-exact path's and the oracle's."""
    orc = _oracle()
    x = clustered(40000, 64, 200, 13, sigma=0.3)
    from vectordb_amd.builder import build_index
    fi = build_index(x, 30, 20, "l2", 6, seed=5)
    assert fi.n_levels() >= 3
    monkeypatch.setenv("CUDDLE_ROUTE_MARGIN", "0")
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 150, 9, noise=0.1)
    lv_g, lc_g = dev.route_batch(q, 8, _ffi.SEARCH_ROUTE_GROUPED)
    lv_e, lc_e = dev.route_batch(q, 8, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    assert np.array_equal(lc_g, lc_e)
    for i in range(q.shape[0]):
        assert np.array_equal(lv_g[i, :lc_g[i]], lv_e[i, :lc_e[i]]), i
    ids, dist, cnt, st = dev.search_batch(q, 10, 8, 5, _ffi.SEARCH_ROUTE_GROUPED, want_stats=True)
    assert st["route_fallbacks"] > q.shape[0] // 2, st
    oi = orc.OracleIndex(fi)
    for i in range(0, q.shape[0], 5):
        t = oi.search(q[i], 10, 8, 5, trace=True)
        assert np.array_equal(ids[i, :cnt[i]], t["ids"][:t["count"]]), i


@pytest.mark.parametrize("metric,d", [("l2", 64), ("cosine", 96), ("dot", 40)])
def test_rabitq_search_matches_oracle(gpu, metric, d):
    """RaBitQ phase 1: rotation, estimates, f32 selection, rerank, top-k — all bit-exact."""
    orc = _oracle()
    x, fi = small_index(n=6000, d=d, n_clusters=40, branching=10, leaf=30, metric=metric, code_kind="rabitq")
    dev = DeviceIndex(fi, gpu)
    oi = orc.OracleIndex(fi)
    q = queries_like(x, 48, 17)
    q[5] = 0.0  # zero query: estimate = norm^2 (rabitq.rs:408-413); cosine distance would be NaN, so L2/dot only
    if metric == "cosine":
        q[5] = x[5]
    _compare_full(fi, dev, q, 10, 5, 10)
    _compare_full(fi, dev, q, 100, 8, 10)
    _compare_full(fi, dev, q, 10, 5, 10, _ffi.SEARCH_STATS_VARIANT)
    leaves, lcnt = dev.route_batch(q, 5)
    pos, key, ccnt = dev.scan_batch(q, leaves, lcnt)
    ids, dist, cnt, spos, skey, sdist, scnt = dev.search_leaves_batch(q, leaves, lcnt, 10, 10)
    for i in range(q.shape[0]):
        t = oi.search(q[i], 10, 5, 10, trace=True)
        assert t["rc"] == 0
        assert np.array_equal(pos[i, :ccnt[i]], t["cand_pos"])
        assert np.array_equal(key[i, :ccnt[i]], t["cand_key"]), i  # f32 bit patterns of the estimates
        assert np.array_equal(spos[i, :scnt[i]], t["sel_pos"])
        assert np.array_equal(skey[i, :scnt[i]], t["sel_key"])
        assert np.array_equal(sdist[i, :scnt[i]].view(np.uint32), t["sel_dist"].view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_mfma_routing_with_massive_ties(gpu, metric):
    """Data with few distinct points: a wide level whose centroids repeat, so most approximate (and
    exact) distances of a query tie. The prefilter's shortlist overflows and falls back to the plain
    selection, the proof fails for tied boundaries and those queries take the exact kernels; the leaf
    lists must still equal the all-exact path's and the oracle's."""
    orc = _oracle()
    rng = np.random.default_rng(17)
    x = rng.integers(0, 2, (60000, 12)).astype(np.float32) + 0.001 * rng.standard_normal((60000, 12)).astype(np.float32)
    x[::3] = np.round(x[::3])  # a third of the points exactly on the 4096 corners
    from vectordb_amd.builder import build_index
    fi = build_index(x, 100, 30, metric, 6, seed=4)
    dev = DeviceIndex(fi, gpu)
    q = np.round(queries_like(x, 120, 8, noise=0.0)).astype(np.float32)  # queries on corners: many exactly equal distances
    for probes in (20, 100):
        lv_mfma, lc_mfma = dev.route_batch(q, probes)
        lv_exact, lc_exact = dev.route_batch(q, probes, _ffi.SEARCH_ROUTE_EXACT_ONLY)
        assert np.array_equal(lc_mfma, lc_exact)
        for i in range(q.shape[0]):
            assert np.array_equal(lv_mfma[i, :lc_mfma[i]], lv_exact[i, :lc_exact[i]]), (probes, i)
        oi = orc.OracleIndex(fi)
        for i in range(0, q.shape[0], 9):
            t = oi.search(q[i], 10, probes, 5, trace=True)
            assert np.array_equal(lv_mfma[i, :lc_mfma[i]], t["leaves"]), (probes, i)
    _compare_full(fi, dev, q[:30], 10, 100, 5)
