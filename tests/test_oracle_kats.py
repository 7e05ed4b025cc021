"""The CPU oracle against every known-answer test the reference's own unit tests hold for the
path (there are no golden files in the reference; these analytic cases are what pins the
restatement). Each test names the reference test it mirrors."""
import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.builder import build_index, make_rotation
from oracle import oracle as orc

L2, COS, DOT = _ffi.METRIC_L2, _ffi.METRIC_COSINE, _ffi.METRIC_DOT


def naive_sum(values):
    s = np.float32(0)
    for v in values:
        s = np.float32(s + np.float32(v))
    return s


# ---- src/distance/simd_neon.rs:351-517 ----------------------------------------------------------
def test_l2_squared_ramp():  # simd_neon.rs:352-371
    a = (np.arange(1024, dtype=np.float32) * np.float32(0.1)).astype(np.float32)
    b = (a + np.float32(0.5)).astype(np.float32)
    expected = naive_sum((a - b) * (a - b))
    assert abs(expected - orc.l2(a, b)) / expected < 1e-4


def test_dot_product_ramp():  # simd_neon.rs:374-393
    a = (np.arange(1024, dtype=np.float32) * np.float32(0.001)).astype(np.float32)
    b = (a + np.float32(0.1)).astype(np.float32)
    expected = naive_sum(a * b)
    assert abs(expected - orc.dot(a, b)) / abs(expected) < 1e-5


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 100, 1023, 1024, 1025])
def test_various_lengths(n):  # simd_neon.rs:396-420
    a = np.arange(n, dtype=np.float32)
    b = a * 2
    expected = naive_sum(a * b)
    got = orc.dot(a, b)
    err = abs(expected - got) / abs(expected) if abs(expected) > 1e-10 else abs(expected - got)
    assert err < 1e-4
    # the values are small integers: up to n = 33 every partial sum is exact, so any order agrees
    if n <= 33:
        assert got == float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        assert orc.l2(a, b) == float(((a - b).astype(np.float64) ** 2).sum())


def test_edge_cases():  # simd_neon.rs:423-438
    e = np.zeros(0, dtype=np.float32)
    assert orc.l2(e, e) == 0.0 and orc.dot(e, e) == 0.0
    assert orc.l2([2.0], [3.0]) == 1.0 and orc.dot([2.0], [3.0]) == 6.0
    v = [1.0, 2.0, 3.0, 4.0]
    assert orc.l2(v, v) == 0.0


def test_l2_batch_matches_single():  # simd_neon.rs:441-491 (n = 4, 3, 6 targets)
    q = np.array([1, 2, 3, 4], dtype=np.float32)
    t = np.array([[1, 2, 3, 4], [2, 3, 4, 5], [0, 0, 0, 0], [5, 5, 5, 5], [1, 1, 1, 1], [2, 2, 2, 2]], dtype=np.float32)
    for n in (4, 3, 6):
        got = orc.l2_batch(q, t[:n])
        assert got.shape == (n,)
        for i in range(n):
            exp = orc.l2(q, t[i])
            assert abs(got[i] - exp) / max(abs(exp), 1e-6) < 1e-5
    assert list(orc.l2_batch(q, t[:4])) == [0.0, 4.0, 30.0, 30.0]


def test_l2_batch_arithmetic_differs_from_single_only_in_rounding():
    rng = np.random.default_rng(5)
    q = rng.standard_normal(259).astype(np.float32)
    t = rng.standard_normal((7, 259)).astype(np.float32)
    got = orc.l2_batch(q, t)
    for i in range(7):
        assert abs(got[i] - orc.l2(q, t[i])) <= 1e-5 * got[i]
    # targets 4..6 go through l2_squared itself (simd_neon.rs:251-254): bit-identical
    for i in range(4, 7):
        assert got[i] == orc.l2(q, t[i])


# ---- tests/integration_tests.rs:3-30, src/distance/mod.rs:103-126 -----------------------------------
def test_distance_calculations():
    a, b = [1.0, 2.0, 3.0], [4.0, 5.0, 6.0]
    assert orc.distance(a, b, L2) == 27.0
    assert orc.distance(a, b, DOT) == -32.0
    c = orc.distance(a, b, COS)
    assert 0.0 <= c <= 2.0 and abs(c - (1 - 32 / np.sqrt(14 * 77))) < 1e-6


def test_identical_vectors_distance():
    v = [1.0, 2.0, 3.0, 4.0]
    assert orc.distance(v, v, L2) == 0.0
    assert abs(orc.distance(v, v, COS)) < 1e-6


def test_batch_distances_identical_is_zero():  # distance/mod.rs:115-126
    assert orc.distance([1.0, 0.0], [1.0, 0.0], L2) == 0.0


# ---- src/quantization/binary.rs:192-291 -----------------------------------------------------------
def test_binary_quantization_bits():  # binary.rs:193-206
    c = orc.binary_quantize([1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 0.1, -0.1], 0.0)
    assert c.shape == (1, 1) and int(c[0, 0]) == 0b01010101
    # strict '>' : a value equal to the threshold gives 0
    assert int(orc.binary_quantize([0.0, 1.0], 0.0)[0, 0]) == 0b10


def test_hamming_kats():  # binary.rs:209-246, :263-276
    q = lambda v: orc.binary_quantize(v, 0.0)[0]  # noqa: E731
    assert orc.hamming(q([1, 1, 1, 1]), q([1, 1, -1, -1])) == 2
    v = np.arange(128, dtype=np.float32)
    assert orc.hamming(q(v), q(v)) == 0
    assert orc.hamming(q(np.ones(64)), q(-np.ones(64))) == 64
    assert orc.hamming(q([1, -1, 1, -1]), q([-1, 1, -1, 1])) == 4


def test_quantizer_from_vectors_mean():  # binary.rs:249-260
    assert abs(orc.binary_threshold([[1, 2, 3], [-1, -2, -3], [0, 0, 0]])) < 1e-6


def test_large_dimension():  # binary.rs:279-291
    v1 = np.arange(1024, dtype=np.float32) - 512
    v2 = np.arange(1024, dtype=np.float32) - 500
    d = orc.hamming(orc.binary_quantize(v1, 0.0)[0], orc.binary_quantize(v2, 0.0)[0])
    assert d == 12  # indices 501..512 differ


def test_threshold_is_sequential_f32():  # binary.rs:127-133: order-dependent single-precision fold
    x = np.full((3000, 128), 100.0, dtype=np.float32)
    x[0, 0] = 2.0 ** 25
    t = orc.binary_threshold(x)
    s = np.float32(0)
    for v in x.reshape(-1)[:2000]:
        s = np.float32(s + v)
    # after the huge first element every +100 is rounded in f32 steps of 4
    assert t == pytest.approx(float(np.cumsum(x.reshape(-1), dtype=np.float32)[-1] / np.float32(x.size)), rel=0, abs=0)
    assert t != float(x.astype(np.float64).mean())


# ---- src/quantization/rabitq.rs:443-527 -----------------------------------------------------------
def test_rabitq_estimate_error_bounds():  # rabitq.rs:444-490
    d = 128
    rot = make_rotation(d, 11)
    rng = np.random.default_rng(2)
    rel = []
    for _ in range(100):
        v1, v2 = rng.random(d, dtype=np.float32), rng.random(d, dtype=np.float32)
        codes, norms, ipoo = orc.rabitq_quantize(rot, v1)
        qnorm, rq = orc.rabitq_query_prep(rot, v2)
        est = orc.rabitq_estimate_fast(codes[0], norms[0], ipoo[0], d, qnorm, rq)
        true = float(((v1 - v2) ** 2).sum())
        rel.append(abs(est - true) / true)
    assert np.mean(rel) < 0.20 and np.max(rel) < 0.70


def test_rabitq_normalization():  # rabitq.rs:493-515
    d = 64
    rot = make_rotation(d, 12)
    v = np.zeros(d, dtype=np.float32)
    v[:2] = [3.0, 4.0]
    codes, norms, ipoo = orc.rabitq_quantize(rot, v)
    assert abs(norms[0] - 5.0) < 1e-5 and 0.0 < ipoo[0] <= d


def test_rabitq_zero_vector():  # rabitq.rs:518-527
    rot = make_rotation(32, 13)
    codes, norms, ipoo = orc.rabitq_quantize(rot, np.zeros(32, dtype=np.float32))
    assert norms[0] == 0.0 and ipoo[0] == 1.0 and not codes.any()
    # estimator special cases, rabitq.rs:342-350
    rq = np.zeros(32, dtype=np.float32)
    assert orc.rabitq_estimate_fast(codes[0], 0.0, 1.0, 32, 0.0, rq) == 0.0
    assert orc.rabitq_estimate_fast(codes[0], 0.0, 1.0, 32, 3.0, rq) == 9.0
    assert orc.rabitq_estimate_fast(codes[0], 2.0, 1.0, 32, 0.0, rq) == 4.0


# ---- src/index/hierarchical.rs:1017-1158 (index-level behaviour on our own deterministic build) -------
def test_two_cluster_search():  # hierarchical.rs:1017-1046
    v = [[i * 0.01, 0.0] for i in range(50)] + [[10.0 + i * 0.01, 10.0] for i in range(50)]
    fi = build_index(np.array(v, dtype=np.float32), 5, 30, "l2", 10, seed=3)
    r = orc.OracleIndex(fi).search([0.25, 0.0], 5, 2, 3)
    assert r["rc"] == 0 and r["count"] == 5 and all(int(i) < 50 for i in r["ids"][:5])


@pytest.mark.parametrize("branching", [3, 5, 10])
def test_branching_factors(branching):  # hierarchical.rs:1090-1109
    x = np.array([[float(i), 0.0] for i in range(200)], dtype=np.float32)
    fi = build_index(x, branching, 50, "l2", 10, seed=4)
    assert fi.n_vectors == 200
    r = orc.OracleIndex(fi).search([100.0, 0.0], 10, 2, 3)
    assert r["count"] == 10


def test_ramp_vectors_search_sorted():  # hierarchical.rs:1112-1158
    x = (np.arange(1000 * 128, dtype=np.float32)).reshape(1000, 128)
    fi = build_index(x, 10, 100, "l2", 20, seed=5)
    r = orc.OracleIndex(fi).search(x[500], 10, 4, 5)
    assert r["count"] == 10
    d = r["dist"][:10]
    assert np.all(d[1:] >= d[:-1]) and d[0] >= 0.0


def test_rabitq_index_search():  # hierarchical_with_rabitq.rs tests (same shapes as the binary ones)
    v = [[i * 0.01, 0.0] for i in range(50)] + [[10.0 + i * 0.01, 10.0] for i in range(50)]
    fi = build_index(np.array(v, dtype=np.float32), 5, 30, "l2", 10, seed=3, code_kind="rabitq")
    r = orc.OracleIndex(fi).search([0.25, 0.0], 5, 2, 3)
    assert r["rc"] == 0 and r["count"] == 5 and all(int(i) < 50 for i in r["ids"][:5])


# ---- error behaviour (hierarchical.rs:337, :651, :402) ---------------------------------------------
def test_error_codes():
    x = np.random.default_rng(0).standard_normal((300, 8)).astype(np.float32)
    oi = orc.OracleIndex(build_index(x, 4, 20, "l2", 5, seed=1))
    assert oi.search(np.zeros(7, np.float32), 5, 2, 3)["rc"] == orc.ORC_ERR_BAD_DIM
    assert oi.search(np.zeros(8, np.float32), 0, 2, 3)["rc"] == orc.ORC_ERR_BAD_ARG
    q = np.zeros(8, np.float32)
    q[0] = np.nan
    assert oi.search(q, 5, 1, 3)["rc"] == orc.ORC_ERR_NAN


def test_kmeans_nearest_centroid():  # kmeans.rs:295-310 with the two centroids a converged fit gives
    cent = np.array([[0.5, 0.5], [10.0, 10.0]], dtype=np.float32)
    a, d = orc.kmeans_assign(np.array([[0.5, 0.5]], dtype=np.float32), cent, _ffi.METRIC_L2)
    assert a[0] == 0 and d[0] == 0.0
    pts = np.array([[1.0, 0.0], [0.0, 1.0], [10.0, 10.0]], dtype=np.float32)
    a, d = orc.kmeans_assign(pts, cent, _ffi.METRIC_L2)
    assert list(a) == [0, 0, 1] and list(d) == [0.5, 0.5, 0.0]


def test_kmeans_assign_two_clusters_and_first_minimum():  # kmeans.rs:256-292; min_by keeps the first of equal minima
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.random((10, 2)) * 0.5, 10.0 + rng.random((10, 2)) * 0.5]).astype(np.float32)
    cent = np.array([[0.25, 0.25], [10.25, 10.25]], dtype=np.float32)
    a, _ = orc.kmeans_assign(pts, cent, _ffi.METRIC_L2)
    assert list(a) == [0] * 10 + [1] * 10
    dup = np.array([[1.0, 1.0], [3.0, 3.0], [1.0, 1.0]], dtype=np.float32)  # centroid 2 duplicates centroid 0
    a, d = orc.kmeans_assign(np.array([[1.0, 1.5], [2.0, 2.0]], dtype=np.float32), dup, _ffi.METRIC_L2)
    assert list(a) == [0, 0] and list(d) == [0.25, 2.0]
    for metric in (_ffi.METRIC_L2, _ffi.METRIC_COSINE, _ffi.METRIC_DOT):  # agrees with distance() pair by pair
        x = rng.standard_normal((50, 37)).astype(np.float32)
        c = rng.standard_normal((9, 37)).astype(np.float32)
        a, d = orc.kmeans_assign(x, c, metric)
        full = np.stack([orc.distance_pairs(x, np.repeat(c[j:j + 1], 50, 0), metric) for j in range(9)], 1)
        assert np.array_equal(a, full.argmin(1)) and np.array_equal(d, full.min(1))


def _oracle_kmeans_fit(x, k, metric, seed, iters=10):
    """KMeans::fit (kmeans.rs:93-133) from the oracle's pieces, random draws from a seeded generator."""
    rng = np.random.default_rng(seed)
    cent, _ = orc.kmeanspp_init(x, k, metric, int(rng.integers(0, len(x))), rng.random(max(k - 1, 1)).astype(np.float32))
    a, _ = orc.kmeans_assign(x, cent, metric)
    prev = a.copy()
    for _ in range(min(iters, 12)):
        cent, counts = orc.kmeans_update(x, a, k, cent)
        for e in np.nonzero(counts == 0)[0]:
            cent[e] = x[int(rng.integers(0, len(x)))]
        a, _ = orc.kmeans_assign(x, cent, metric)
        if (a != prev).mean() < 0.02:
            break
        prev = a.copy()
    return cent, a


@pytest.mark.parametrize("seed", range(5))
def test_kmeans_simple_two_clusters(seed):  # kmeans.rs:256-292
    rng = np.random.default_rng(100 + seed)
    x = np.concatenate([rng.random((10, 2)) * 0.5, 10.0 + rng.random((10, 2)) * 0.5]).astype(np.float32)
    cent, a = _oracle_kmeans_fit(x, 2, _ffi.METRIC_L2, seed)
    assert len(a) == 20 and len(set(a[:10])) == 1 and len(set(a[10:])) == 1 and a[0] != a[10]


def test_kmeans_get_clusters_and_nearest():  # kmeans.rs:295-328
    x = np.array([[1.0, 0.0], [1.1, 0.1], [10.0, 10.0], [10.1, 10.1]], dtype=np.float32)
    cent, a = _oracle_kmeans_fit(x, 2, _ffi.METRIC_L2, 3)
    assert sorted(np.bincount(a, minlength=2).tolist()) == [2, 2]
    x3 = np.array([[1.0, 0.0], [0.0, 1.0], [10.0, 10.0]], dtype=np.float32)
    cent, _ = _oracle_kmeans_fit(x3, 2, _ffi.METRIC_L2, 1)
    near, _ = orc.kmeans_assign(np.array([[0.5, 0.5]], dtype=np.float32), cent, _ffi.METRIC_L2)
    assert cent[near[0]][0] < 5.0 and cent[near[0]][1] < 5.0


def test_kmeans_update_is_the_mean_and_keeps_empty_rows():  # kmeans.rs:204-215
    x = np.array([[1, 2], [3, 4], [5, 6], [100, 100]], dtype=np.float32)
    prev = np.full((3, 2), -7.0, dtype=np.float32)
    cent, counts = orc.kmeans_update(x, np.array([0, 0, 0, 2]), 3, prev)
    assert counts.tolist() == [3, 0, 1]
    assert cent.tolist() == [[3.0, 4.0], [-7.0, -7.0], [100.0, 100.0]]


def test_kmeanspp_walk_picks_by_cumulative_weight():  # kmeans.rs:66-79
    x = np.array([[0.0], [1.0], [2.0], [10.0]], dtype=np.float32)
    # first = 0: L2 distances 0, 1, 4, 100 -> weights 0, 1, 16, 10000 (total 10017)
    for u, want in [(0.0, 0), (0.00005, 1), (0.001, 2), (0.5, 3), (0.9999, 3)]:
        _, chosen = orc.kmeanspp_init(x, 2, _ffi.METRIC_L2, 0, np.array([u], dtype=np.float32))
        assert chosen.tolist() == [0, want], (u, chosen)
