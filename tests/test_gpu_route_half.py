"""GPU: the single-pass half-precision routing filter (a level every query sees whole). Its leaf lists must equal
the split-bf16 filter's and the all-exact kernels' whatever the magnitude of the data, its measured error must stay
inside the bound its proof uses, and where the proof cannot hold (a margin of one) the exact path must take over."""
import os
import zlib

import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
from vectordb_amd.index import DeviceIndex

pytestmark = pytest.mark.gpu


def _two_level(rng, d, roots, fan, metric, cent_fn):
    """roots -> roots * fan leaves of two vectors each; the centroids of the wide level come from cent_fn."""
    n_nodes = roots + roots * fan
    cent = np.zeros((n_nodes, d), dtype=np.float32)
    cent[:roots] = cent_fn(roots)
    cent[roots:] = cent_fn(roots * fan)
    child_offsets = np.zeros(n_nodes + 1, dtype=np.uint32)
    child_offsets[1:roots + 1] = np.arange(1, roots + 1) * fan
    child_offsets[roots + 1:] = roots * fan
    children = np.arange(roots, n_nodes, dtype=np.uint32)
    vec_offsets = np.zeros(n_nodes + 1, dtype=np.uint64)
    vec_offsets[roots + 1:] = np.arange(1, roots * fan + 1) * 2
    n = roots * fan * 2
    x = np.repeat(cent[roots:], 2, axis=0).astype(np.float32)
    return FlatIndex(dim=d, metric=metric, code_kind=_ffi.CODE_BINARY, centroids=cent, child_offsets=child_offsets,
                     children=children, roots=np.arange(roots, dtype=np.uint32), vec_offsets=vec_offsets,
                     ids=np.arange(n, dtype=np.uint64), vectors=x, codes=pack_binary_codes(x, 0.0), threshold=0.0)


def _lists(dev, q, probes, flags=0):
    lv, lc = dev.route_batch(q, probes, flags)
    return [lv[i, :lc[i]].tolist() for i in range(q.shape[0])]


CASES = {
    "plain": lambda rng, d: (lambda m: rng.standard_normal((m, d)).astype(np.float32)),
    "far_from_origin": lambda rng, d: (lambda m: (rng.standard_normal((m, d)) + 1000.0).astype(np.float32)),
    "tiny": lambda rng, d: (lambda m: (rng.standard_normal((m, d)) * 1e-12).astype(np.float32)),
    "huge": lambda rng, d: (lambda m: (rng.standard_normal((m, d)) * 1e12).astype(np.float32)),
    # rows whose norms differ by ten orders of magnitude, components that differ by five inside a row
    "mixed": lambda rng, d: (lambda m: (rng.standard_normal((m, d)) * 10.0 ** rng.integers(-5, 6, (m, 1))
                                        * 10.0 ** rng.integers(-5, 1, (m, d))).astype(np.float32)),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("metric", [_ffi.METRIC_L2, _ffi.METRIC_DOT, _ffi.METRIC_COSINE])
@pytest.mark.parametrize("d", [24, 130])
def test_half_filter_lists_equal_split_and_exact(gpu, metric, d, case):
    rng = np.random.default_rng(zlib.crc32(f"{metric}/{d}/{case}".encode()))
    gen = CASES[case](rng, d)
    fi = _two_level(rng, d, 40, 75, metric, gen)  # level 1: 3000 nodes, whole for every query (40 roots <= probes)
    if metric != _ffi.METRIC_COSINE:  # (the cosine distance to a zero vector is NaN, on which the reference panics)
        fi.centroids[45] = 0.0
        fi.centroids[1045] = 0.0
    dev = DeviceIndex(fi, gpu)
    q = gen(64)
    for probes in (40, 100):
        want = _lists(dev, q, probes, _ffi.SEARCH_ROUTE_EXACT_ONLY)
        assert _lists(dev, q, probes) == want
        assert _lists(dev, q, probes, _ffi.SEARCH_ROUTE_SPLIT_FILTER) == want
    ratio, _ = dev.route_filter_error(q, 100, 1)
    assert ratio < 1.0, ratio  # the proof's bound holds (it is the half-precision filter that is measured)


def test_half_filter_error_is_the_half_filters(gpu):
    """The diagnostic measures the filter a search uses: with rounding to 11 bits the error is far above what the
    split operands leave, and still well inside its own bound."""
    rng = np.random.default_rng(5)
    gen = CASES["plain"](rng, 64)
    dev = DeviceIndex(_two_level(rng, 64, 40, 75, _ffi.METRIC_L2, gen), gpu)
    q = gen(32)
    ratio, err = dev.route_filter_error(q, 100, 1)
    assert 1e-4 < ratio < 0.5, (ratio, err)


def test_margin_of_one_falls_back_to_the_exact_kernels(gpu):
    rng = np.random.default_rng(9)
    gen = CASES["plain"](rng, 24)
    dev = DeviceIndex(_two_level(rng, 24, 40, 75, _ffi.METRIC_L2, gen), gpu)
    q = gen(64)
    want = _lists(dev, q, 100, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    os.environ["CUDDLE_ROUTE_MARGIN_HALF"] = "1"
    try:
        _, _, _, st = dev.search_batch(q, 10, 100, 5, want_stats=True)  # (the handle's first call: later ones may have learnt)
        got = _lists(dev, q, 100)
    finally:
        del os.environ["CUDDLE_ROUTE_MARGIN_HALF"]
    assert got == want
    assert st["route_fallbacks"] > 0, st  # one spare candidate cannot absorb the filter's error for every query


def test_a_level_whose_proofs_keep_failing_returns_to_the_split_filter(gpu):
    """With one spare candidate the half-precision proofs fail for many queries; the failures are counted per level,
    and after the first calls the level is routed with the split filter again: exact lists throughout, and the
    fallbacks of a late search are those of the split filter (a margin of 24: next to none)."""
    rng = np.random.default_rng(13)
    gen = CASES["plain"](rng, 24)
    dev = DeviceIndex(_two_level(rng, 24, 40, 75, _ffi.METRIC_L2, gen), gpu)
    q = gen(128)
    want = _lists(dev, q, 100, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    os.environ["CUDDLE_ROUTE_MARGIN_HALF"] = "1"
    try:
        _, _, _, first = dev.search_batch(q, 10, 100, 5, want_stats=True)
        for _ in range(6):
            assert _lists(dev, q, 100) == want
        _, _, _, late = dev.search_batch(q, 10, 100, 5, want_stats=True)
    finally:
        del os.environ["CUDDLE_ROUTE_MARGIN_HALF"]
    assert first["route_fallbacks"] >= 8, first
    assert late["route_fallbacks"] * 4 <= first["route_fallbacks"], (first, late)


@pytest.mark.parametrize("metric", [_ffi.METRIC_L2, _ffi.METRIC_DOT, _ffi.METRIC_COSINE])
def test_half_filter_on_per_query_frontiers(gpu, metric):
    """A level of per-query frontiers (the 100 x 70 children of the chosen nodes) is contracted from half-precision
    operands when its table is too large for the Infinity Cache; with that size set to zero a small tree takes the
    path: the lists must be the split filter's and the exact kernels'."""
    import torch

    from vectordb_amd.synthetic import build_tree_index_on_device

    os.environ["CUDDLE_GROUPED_HALF_MIN_BYTES"] = "0"
    try:
        dti, queries = build_tree_index_on_device(40, 70, 3, 3, torch.device("cuda", gpu), seed=21, n_queries=96, metric=metric,
                                                  style="gauss")
        dev = dti.create(gpu)
        q = queries.cpu().numpy()
        want = _lists(dev, q, 100, _ffi.SEARCH_ROUTE_EXACT_ONLY)
        got = _lists(dev, q, 100)
        got_split = _lists(dev, q, 100, _ffi.SEARCH_ROUTE_SPLIT_FILTER)
        ids, dist, cnt, st = dev.search_batch(q, 10, 100, 5, want_stats=True)
        ids_x, dist_x, cnt_x = dev.search_batch(q, 10, 100, 5, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    finally:
        del os.environ["CUDDLE_GROUPED_HALF_MIN_BYTES"]
    assert got == want
    assert got_split == want
    assert np.array_equal(ids, ids_x) and np.array_equal(dist.view(np.uint32), dist_x.view(np.uint32)) and np.array_equal(cnt, cnt_x)


def test_failed_half_proofs_on_per_query_frontiers_take_the_exact_path(gpu):
    """The same level with one spare candidate: most proofs fail, the queries are re-routed by the exact kernels from
    their segment tables (lists still exact), and the level then returns to the split filter."""
    import torch

    from vectordb_amd.synthetic import build_tree_index_on_device

    os.environ["CUDDLE_GROUPED_HALF_MIN_BYTES"] = "0"
    os.environ["CUDDLE_ROUTE_MARGIN_HALF"] = "1"
    try:
        dti, queries = build_tree_index_on_device(40, 70, 3, 3, torch.device("cuda", gpu), seed=22, n_queries=128,
                                                  metric=_ffi.METRIC_L2, style="gauss")
        dev = dti.create(gpu)
        q = queries.cpu().numpy()
        _, _, _, first = dev.search_batch(q, 10, 100, 5, want_stats=True)
        want = _lists(dev, q, 100, _ffi.SEARCH_ROUTE_EXACT_ONLY)
        for _ in range(5):
            assert _lists(dev, q, 100) == want
        ids, dist, cnt, late = dev.search_batch(q, 10, 100, 5, want_stats=True)
        ids_x, dist_x, cnt_x = dev.search_batch(q, 10, 100, 5, _ffi.SEARCH_ROUTE_EXACT_ONLY)
    finally:
        del os.environ["CUDDLE_GROUPED_HALF_MIN_BYTES"]
        del os.environ["CUDDLE_ROUTE_MARGIN_HALF"]
    assert first["route_fallbacks"] >= 8, first
    assert late["route_fallbacks"] * 4 <= first["route_fallbacks"], (first, late)
    assert np.array_equal(ids, ids_x) and np.array_equal(dist.view(np.uint32), dist_x.view(np.uint32)) and np.array_equal(cnt, cnt_x)


def test_a_level_table_beyond_the_infinity_cache_takes_half_operands_by_itself(gpu):
    """No size override: a last level of 81^3 = 531 441 nodes of 128-d rows has a 272 MB split table, more than
    the 256 MiB Infinity Cache, so its per-query frontiers are contracted from half-precision operands by the
    library's own rule. The results must be the all-exact path's on every query and the CPU oracle's on a sample
    (the oracle searches a host copy that holds the whole tree and the leaves these queries reach); that the rule
    fired is shown on a second handle with one spare candidate, where half-precision proofs fail in numbers (the
    split filter does not read that setting)."""
    import torch

    import bench
    from vectordb_amd.synthetic import build_tree_index_on_device

    assert "CUDDLE_GROUPED_HALF_MIN_BYTES" not in os.environ
    device = torch.device("cuda", gpu)
    dti, queries = build_tree_index_on_device(128, 81, 3, 4, device, seed=0xBEEF, n_queries=256, metric=_ffi.METRIC_L2,
                                              style="sift")
    n_last = 81 ** 3
    assert n_last * 128 * 4 > 256 << 20
    k, probes, rf = 10, 100, 10
    dev = dti.create(gpu)
    nq = queries.shape[0]
    out = (torch.empty((nq, k), dtype=torch.int64, device=device), torch.empty((nq, k), dtype=torch.float32, device=device),
           torch.empty((nq,), dtype=torch.int32, device=device))
    exact = tuple(torch.empty_like(t) for t in out)
    for _ in range(3):  # (the first calls are the ones the failure feedback has not judged yet: every one must be exact)
        dev.search_batch_device(queries, k, probes, rf, 0, *out)
        torch.cuda.synchronize()
        dev.search_batch_device(queries, k, probes, rf, _ffi.SEARCH_ROUTE_EXACT_ONLY, *exact)
        torch.cuda.synchronize()
        assert torch.equal(out[0], exact[0]) and torch.equal(out[1].view(torch.int32), exact[1].view(torch.int32))
        assert torch.equal(out[2], exact[2])
    chk = bench.oracle_spot_check(dti, dev, queries, k, probes, rf, *out, n_sample=32)
    assert chk["oracle_rc"] == 0 and chk["gpu_ids_identical_on_sample"] and chk["gpu_dist_bits_identical_on_sample"], chk
    os.environ["CUDDLE_ROUTE_MARGIN_HALF"] = "1"
    try:
        other = dti.create(gpu)
        q = queries.cpu().numpy()
        _, _, _, first = other.search_batch(q, k, probes, rf, want_stats=True)
    finally:
        del os.environ["CUDDLE_ROUTE_MARGIN_HALF"]
    assert first["route_fallbacks"] >= 8, first
