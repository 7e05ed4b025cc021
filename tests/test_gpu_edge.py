"""GPU edge cases through the C ABI: leaf helpers on all tail lengths, error codes, empty and
ragged inputs, shards with empty leaves, tie-heavy selections, the 10 000 rerank cap, the
global spill path of the scan (more candidates than fit in LDS) and batch chunking."""
import ctypes as C

import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.builder import build_index
from vectordb_amd.flat_index import FlatIndex
from vectordb_amd.index import ClusteredIndex, DeviceIndex

from helpers import clustered, queries_like, small_index
from test_gpu_parity import _compare_full, _oracle

pytestmark = pytest.mark.gpu
L2, COS, DOT = _ffi.METRIC_L2, _ffi.METRIC_COSINE, _ffi.METRIC_DOT


def _pairs(a, b, metric):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros(a.shape[0], dtype=np.float32)
    _ffi.check(_ffi.lib().cuddle_distance_pairs(_ffi.ptr(a), _ffi.ptr(b), a.shape[0], a.shape[1], metric, _ffi.ptr(out), 0))
    return out


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 100, 1023, 1024, 1025])
def test_distance_all_lengths_bit_exact(gpu, n):  # simd_neon.rs:396-420 lengths, all three metrics
    orc = _oracle()
    rng = np.random.default_rng(n)
    a = rng.standard_normal((37, n)).astype(np.float32)
    b = rng.standard_normal((37, n)).astype(np.float32)
    for metric in (L2, COS, DOT):
        got = _pairs(a, b, metric)
        exp = orc.distance_pairs(a, b, metric)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (n, metric)


def test_distance_kats(gpu):  # simd_neon.rs:423-438, tests/integration_tests.rs:3-30
    assert _pairs([[2.0]], [[3.0]], L2)[0] == 1.0 and _pairs([[2.0]], [[3.0]], DOT)[0] == -6.0
    v = [[1.0, 2.0, 3.0, 4.0]]
    assert _pairs(v, v, L2)[0] == 0.0 and abs(_pairs(v, v, COS)[0]) < 1e-6
    a, b = [[1.0, 2.0, 3.0]], [[4.0, 5.0, 6.0]]
    assert _pairs(a, b, L2)[0] == 27.0 and _pairs(a, b, DOT)[0] == -32.0 and 0.0 <= _pairs(a, b, COS)[0] <= 2.0


def test_quantize_and_hamming_kats(gpu):  # binary.rs:193-291
    orc = _oracle()
    rng = np.random.default_rng(0)
    for d in (4, 8, 63, 64, 65, 128, 960, 1024):
        v = rng.standard_normal((50, d)).astype(np.float32)
        codes = np.zeros((50, (d + 63) // 64), dtype=np.uint64)
        _ffi.check(_ffi.lib().cuddle_binary_quantize(_ffi.ptr(v), 50, d, 0.0, _ffi.ptr(codes), 0))
        assert np.array_equal(codes, orc.binary_quantize(v, 0.0))
        out = np.zeros(25, dtype=np.uint32)
        a, b = np.ascontiguousarray(codes[:25]), np.ascontiguousarray(codes[25:])
        _ffi.check(_ffi.lib().cuddle_hamming_pairs(_ffi.ptr(a), _ffi.ptr(b), 25, codes.shape[1], _ffi.ptr(out), 0))
        assert np.array_equal(out, orc.hamming_pairs(a, b))
    one = np.ones((1, 64), dtype=np.float32)
    c1 = np.zeros((1, 1), dtype=np.uint64)
    c2 = np.zeros((1, 1), dtype=np.uint64)
    _ffi.lib().cuddle_binary_quantize(_ffi.ptr(one), 1, 64, 0.0, _ffi.ptr(c1), 0)
    _ffi.lib().cuddle_binary_quantize(_ffi.ptr(-one), 1, 64, 0.0, _ffi.ptr(c2), 0)
    out = np.zeros(1, dtype=np.uint32)
    _ffi.lib().cuddle_hamming_pairs(_ffi.ptr(c1), _ffi.ptr(c2), 1, 1, _ffi.ptr(out), 0)
    assert out[0] == 64  # binary.rs:235-246


def test_error_codes(gpu):
    x, fi = small_index(n=600, d=8, n_clusters=6, branching=4, leaf=20)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 4, 1)
    with pytest.raises(_ffi.CuddleError) as e:
        dev.search_batch(q[:, :7], 5, 2, 3)
    assert e.value.status == _ffi.ERR_BAD_DIM            # hierarchical.rs:337
    for k, p, rf in ((0, 2, 3), (5, 0, 3), (5, 2, 0)):
        with pytest.raises(_ffi.CuddleError) as e:
            dev.search_batch(q, k, p, rf)
        assert e.value.status == _ffi.ERR_BAD_ARG        # k-1 underflow, hierarchical.rs:651
    _compare_full(fi, dev, q, 100, 2, 200, _ffi.SEARCH_STATS_VARIANT)  # uncapped 20000 > all 600 vectors: all of them
    ids, dist, cnt = dev.search_batch(q[:0], 5, 2, 3)    # empty batch
    assert ids.shape == (0, 5)
    # the handle still works after errors
    _compare_full(fi, dev, q, 5, 2, 3)


def test_fewer_results_than_k_and_single_leaf(gpu):
    x = clustered(40, 12, 2, 3)
    fi = build_index(x, 4, 30, "l2", 5, seed=1)          # 40 <= 2*30 -> one leaf, which is the only root
    assert fi.n_nodes == 1
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 6, 2)
    _compare_full(fi, dev, q, 100, 3, 2)                  # at most 40 results
    ids, dist, cnt = dev.search_batch(q, 100, 3, 2)
    assert np.all(cnt == 40)
    ci = ClusteredIndex(fi, gpu)
    r = ci.search(q[0], 5, 1, 1)
    assert len(r) == 5 and all(r[i][1] <= r[i + 1][1] for i in range(4))
    r2, st = ci.search_with_stats(q[0], 5, 1, 1)
    assert st.total_leaves == 1 and st.leaves_searched == 1 and st.vectors_scanned_binary == 40
    assert st.leaves_searched_ids == [0] and st.probes_per_level == [1, 0]


def test_empty_leaves_and_shards(gpu):
    orc = _oracle()
    x, fi = small_index(n=3000, d=24, n_clusters=15, branching=6, leaf=25)
    q = queries_like(x, 32, 4)
    for world in (2, 5):
        for rank in range(world):
            sh = fi.shard(rank, world)
            _compare_full(sh, DeviceIndex(sh, gpu), q, 10, 4, 4)
    # a shard with no vectors at all: every query returns nothing (hierarchical.rs:452-454)
    empty = fi.shard(0, 1)
    empty = FlatIndex(dim=fi.dim, metric=fi.metric, code_kind=fi.code_kind, centroids=fi.centroids,
                      child_offsets=fi.child_offsets, children=fi.children, roots=fi.roots,
                      vec_offsets=np.zeros_like(fi.vec_offsets), ids=fi.ids[:0], vectors=fi.vectors[:0],
                      threshold=fi.threshold, codes=fi.codes[:0])
    ids, dist, cnt = DeviceIndex(empty, gpu).search_batch(q, 10, 4, 4)
    assert np.all(cnt == 0) and np.all(ids == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isinf(dist))


def test_tie_heavy_selection(gpu):
    """8-bit codes on integer data: almost every Hamming value is shared by many candidates, so
    the selection boundary is always inside a tie class (scan-order rule)."""
    rng = np.random.default_rng(5)
    x = rng.integers(0, 4, (4000, 8)).astype(np.float32)
    fi = build_index(x, 8, 40, "l2", 6, seed=2)
    dev = DeviceIndex(fi, gpu)
    q = rng.integers(0, 4, (40, 8)).astype(np.float32)
    _compare_full(fi, dev, q, 10, 6, 3)
    _compare_full(fi, dev, q, 3, 10, 7)


@pytest.mark.parametrize("code_kind", ["binary", "rabitq"])
def test_top_k_of_selections_between_1024_and_2048(gpu, code_kind):
    """k = 10 of 1500 and of 2040 reranked candidates: the one-wave top-k with 32 keys per lane (duplicates in the data,
    so that equal distances are ordered by the phase-1 order)."""
    x = clustered(30000, 24, 60, 12, sigma=0.3)
    x[1000:1100] = x[2000:2100]
    fi = build_index(x, 20, 60, "l2", 6, seed=6, code_kind=code_kind)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 24, 8)
    q[0] = x[1000]
    _compare_full(fi, dev, q, 10, 60, 150)
    _compare_full(fi, dev, q, 10, 60, 204)


def test_rerank_cap_and_spill_path(gpu):
    """probes large enough that one query scans > 24576 candidates (the LDS capacity of the scan
    kernel: distances go to the global spill buffer) and k*rerank_factor hits the 10 000 cap."""
    x = clustered(70000, 32, 200, 8, sigma=0.3)
    fi = build_index(x, 100, 50, "l2", 6, seed=4)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 6, 6)
    ids, dist, cnt, st = dev.search_batch(q, 100, 800, 150, want_stats=True)
    assert st["max_candidates"] > 24576, st
    assert st["vectors_reranked"] == 6 * 10000          # min(k*rf, 10000), hierarchical.rs:447
    _compare_full(fi, dev, q, 100, 800, 150)
    with pytest.raises(_ffi.CuddleError) as e:           # uncapped variant: 20000 of > 24576 reachable candidates
        dev.search_batch(q, 100, 800, 200, _ffi.SEARCH_STATS_VARIANT)
    assert e.value.status == _ffi.ERR_UNSUPPORTED


def test_batch_larger_than_one_chunk(gpu):
    x, fi = small_index(n=3000, d=16, n_clusters=20, branching=6, leaf=30)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 5000, 21)                        # > 4096: two chunks
    ids, dist, cnt = dev.search_batch(q, 5, 3, 4)
    sel = np.r_[0:40, 4090:4110, 4960:5000]
    ids2, dist2, cnt2 = dev.search_batch(q[sel], 5, 3, 4)
    assert np.array_equal(ids[sel], ids2) and np.array_equal(dist[sel], dist2)
    orc = _oracle()
    oi = orc.OracleIndex(fi)
    for i in (0, 2499, 2500, 4095, 4096, 4999):  # (the two pieces of such a batch run on two streams: 0..2499, 2500..4999)
        r = oi.search(q[i], 5, 3, 4)
        assert np.array_equal(ids[i], r["ids"]) and np.array_equal(dist[i], r["dist"])
    # the statistics of a batch that ran in pieces are those of its parts
    _, _, _, whole = dev.search_batch(q, 5, 3, 4, want_stats=True)
    parts = [dev.search_batch(q[a:b], 5, 3, 4, want_stats=True)[3] for a, b in ((0, 3000), (3000, 5000))]
    for key in ("leaves_searched", "vectors_scanned", "vectors_reranked"):
        assert whole[key] == parts[0][key] + parts[1][key], key
    assert whole["max_candidates"] == max(p["max_candidates"] for p in parts)


@pytest.mark.parametrize("code_kind", ["binary", "rabitq"])
def test_four_pieces_on_two_streams(gpu, code_kind):
    """9000 queries = four pieces of 2250 alternating between the handle's two streams and work-space halves; the
    device-pointer entry point on a stream of the caller's, twice in a row (the second call must wait for the first's
    pieces), against single-chunk searches of slices."""
    import torch
    x, fi = small_index(n=3000, d=24, n_clusters=20, branching=6, leaf=30, code_kind=code_kind)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 9000, 5)
    dq = torch.from_numpy(q).cuda(gpu)
    st = torch.cuda.Stream(gpu)
    st.wait_stream(torch.cuda.current_stream(gpu))
    a = dev.search_batch_device(dq, 10, 3, 4, stream=st.cuda_stream)
    b = dev.search_batch_device(dq.flip(0).contiguous(), 10, 3, 4, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ids_a, ids_b = a[0].cpu().numpy(), b[0].cpu().numpy()
    assert np.array_equal(ids_a, ids_b[::-1])
    for lo in (0, 2200, 4400, 8000):
        ids, dist, cnt = dev.search_batch(q[lo:lo + 1000], 10, 3, 4)
        assert np.array_equal(ids.view(np.int64), ids_a[lo:lo + 1000])
        assert np.array_equal(dist.view(np.uint32), a[1].cpu().numpy()[lo:lo + 1000].view(np.uint32))


def test_device_pointer_entry_point_matches_host_one(gpu):
    import torch

    x, fi = small_index(n=3000, d=16, n_clusters=20, branching=6, leaf=30)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 100, 2)
    ids, dist, cnt = dev.search_batch(q, 10, 3, 4)
    dq = torch.from_numpy(q).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tid, tdist, tcnt = dev.search_batch_device(dq, 10, 3, 4, _ffi.SEARCH_TIME_STAGES)
    s.synchronize()
    assert np.array_equal(tid.cpu().numpy().view(np.uint64), ids) and np.array_equal(tdist.cpu().numpy(), dist)
    t = dev.stage_times()
    assert t["searches"] == 1 and t["scan"] > 0.0


def test_device_array_index_matches_oracle(gpu):
    """Index built on the GPU and adopted in place (no host copy) gives the oracle's answers."""
    import torch
    from vectordb_amd.synthetic import build_tree_index_on_device

    orc = _oracle()
    dti, q = build_tree_index_on_device(dim=32, fanout=6, levels=3, leaf_size=40, device="cuda:0", seed=3, n_queries=50)
    dev = dti.create(gpu)
    fi = dti.to_flat()
    assert fi.n_vectors == 6 ** 3 * 40
    ids, dist, cnt = dev.search_batch(q.cpu().numpy(), 10, 4, 6)
    oi = orc.OracleIndex(fi)
    for i in range(q.shape[0]):
        r = oi.search(q[i].cpu().numpy(), 10, 4, 6)
        assert np.array_equal(ids[i], r["ids"]) and np.array_equal(dist[i].view(np.uint32), r["dist"].view(np.uint32)), i


def test_device_array_rabitq_index_matches_oracle(gpu):
    """The same with RaBitQ codes encoded on the GPU, cosine metric."""
    from vectordb_amd.synthetic import build_tree_index_on_device

    orc = _oracle()
    dti, q = build_tree_index_on_device(dim=48, fanout=5, levels=3, leaf_size=30, device="cuda:0", seed=4, n_queries=40,
                                        metric=_ffi.METRIC_COSINE, style="gauss", code_kind=_ffi.CODE_RABITQ)
    dev = dti.create(gpu)
    fi = dti.to_flat()
    assert fi.n_vectors == 5 ** 3 * 30 and fi.code_kind == _ffi.CODE_RABITQ
    ids, dist, cnt = dev.search_batch(q.cpu().numpy(), 10, 3, 5)
    oi = orc.OracleIndex(fi)
    for i in range(q.shape[0]):
        r = oi.search(q[i].cpu().numpy(), 10, 3, 5)
        assert np.array_equal(ids[i], r["ids"]) and np.array_equal(dist[i].view(np.uint32), r["dist"].view(np.uint32)), i


def test_leaf_size_limit(gpu):
    """The scan's tile tables hold the tile number inside a leaf in 12 bits: a leaf of 4096 * 64 vectors is the
    largest that is searched (here a one-leaf index of exactly that size, checked against brute force), one more
    vector is refused at cuddle_index_create."""
    from vectordb_amd import _ffi
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes

    d = 8
    rng = np.random.default_rng(3)

    def one_leaf(n):
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x, FlatIndex(dim=d, metric=_ffi.METRIC_L2, code_kind=_ffi.CODE_BINARY, centroids=x.mean(0, keepdims=True).astype(np.float32),
                            child_offsets=np.array([0, 0], np.uint32), children=np.zeros(0, np.uint32), roots=np.array([0], np.uint32),
                            vec_offsets=np.array([0, n], np.uint64), ids=np.arange(n, dtype=np.uint64), vectors=x,
                            codes=pack_binary_codes(x, 0.0), threshold=0.0)

    x, fi = one_leaf(4096 * 64)
    dev = DeviceIndex(fi, gpu)
    q = x[[5, 70000, 262143]] + np.float32(0.001)
    ids, dist, cnt = dev.search_batch(q, 3, 1, 100000)  # rerank budget (capped at 10000) >> k: the true neighbours survive phase 1
    assert cnt.tolist() == [3, 3, 3] and ids[:, 0].tolist() == [5, 70000, 262143]
    _, big = one_leaf(4096 * 64 + 1)
    with pytest.raises(_ffi.CuddleError) as e:
        DeviceIndex(big, gpu)
    assert e.value.status == _ffi.ERR_UNSUPPORTED


def test_dimension_beyond_16384(gpu):
    """A query vector of more than 64 KiB: the kernels that stage it in LDS (routing distances, refine, rerank)
    need the raised dynamic-LDS limit. Same results as the oracle, bit for bit."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((300, 17000)).astype(np.float32)
    fi = build_index(x, 4, 40, "l2", 4, seed=1)
    dev = DeviceIndex(fi, gpu)
    q = (x[:5] + 0.05 * rng.standard_normal((5, 17000))).astype(np.float32)
    _compare_full(fi, dev, q, 5, 3, 4)


def test_selection_larger_than_the_index(gpu):
    """search_with_stats has no cap on k * rerank_factor (hierarchical.rs:831-938): a selection far beyond the
    candidates a query can reach means all of them, not an error."""
    from vectordb_amd import _ffi
    for kind in ("binary", "rabitq"):
        x, fi = small_index(n=3000, d=24, n_clusters=15, branching=6, leaf=25, code_kind=kind)
        dev = DeviceIndex(fi, gpu)
        q = queries_like(x, 5, 4)
        _compare_full(fi, dev, q, 3000, 6, 1000, _ffi.SEARCH_STATS_VARIANT)
        _compare_full(fi, dev, q, 3000, 1000, 1000, _ffi.SEARCH_STATS_VARIANT)


@pytest.mark.parametrize("code_kind", ["binary", "rabitq"])
@pytest.mark.parametrize("nq", [1, 7, 13, 203])
def test_block_orders_and_lds_sizes_give_the_same_answers(gpu, monkeypatch, code_kind, nq):
    """The order in which the scan and the rerank take the queries (by work, or by the place of each query's first leaf
    with an eighth of that order per XCD: batches that are not multiples of 8 leave places empty) and the size of the
    scan's LDS distance array (queries beyond it keep their distances in global memory) are performance choices: every
    combination must give the oracle's answers."""
    x, fi = small_index(n=4000, d=24, n_clusters=30, branching=6, leaf=25, code_kind=code_kind)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, nq, 17)
    base = dev.search_batch(q, 8, 5, 6)
    oi = _oracle().OracleIndex(fi)
    for i in range(0, nq, max(1, nq // 9)):
        r = oi.search(q[i], 8, 5, 6)
        assert np.array_equal(base[0][i, :r["count"]], r["ids"][:r["count"]]) and base[2][i] == r["count"]
    for env in ({"CUDDLE_SCAN_ORDER": "1"}, {"CUDDLE_SCAN_ORDER": "0"}, {"CUDDLE_LDS_DIST_BYTES": "300"},
                {"CUDDLE_SCAN_ORDER": "1", "CUDDLE_LDS_DIST_BYTES": "300"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = DeviceIndex(fi, gpu).search_batch(q, 8, 5, 6)
        for k in env:
            monkeypatch.delenv(k)
        assert all(np.array_equal(a, b) for a, b in zip(base, got)), env
