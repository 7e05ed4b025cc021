"""GPU: the exact leaf-sharded protocol (cuddle_shard_scan / _ties / _finish + ordered merge) must
reproduce the unsharded search bit for bit, whatever the number of shards — here the shards are
separate handles on the one GPU and the two all-reduces are in-process sums."""
import numpy as np
import pytest
import torch

from vectordb_amd.index import DeviceIndex
from vectordb_amd.sharded import ExactShardedSearcher, exact_search_in_process

from helpers import clustered, queries_like, small_index

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _check(fi, gpu, q, cases, worlds=(2, 3)):
    full = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(q).cuda(gpu)
    for world in worlds:
        shards = [DeviceIndex(fi.shard(r, world), gpu) for r in range(world)]
        for (k, probes, rf) in cases:
            ids0, d0, c0 = full.search_batch_device(dq, k, probes, rf)
            # every shard routes the batch / each routes a slice of it / the same, written into message buffers
            for shard_routing, packed in ((False, False), (True, False), (True, True), (False, True)):
                ids1, d1, c1 = exact_search_in_process(shards, dq, k, probes, rf, shard_routing=shard_routing, packed=packed)
                torch.cuda.synchronize()
                assert torch.equal(c0.cpu(), c1.cpu()), (world, k, probes, rf, shard_routing, packed)
                assert torch.equal(ids0.cpu(), ids1.cpu()), (world, k, probes, rf, shard_routing, packed)
                assert np.array_equal(_bits(d0.cpu().numpy()), _bits(d1.cpu().numpy())), (world, k, probes, rf, shard_routing, packed)


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_sharded_equals_unsharded(gpu, metric):
    x, fi = small_index(n=6000, d=48, n_clusters=40, branching=8, leaf=40, metric=metric)
    q = queries_like(x, 48, 21)
    _check(fi, gpu, q, [(10, 4, 3), (10, 6, 10), (1, 2, 1), (20, 8, 1000), (100, 8, 2)])


def test_sharded_tie_heavy_and_wide_codes(gpu):
    # few distinct Hamming values -> the tie budget decides most of the selection
    rng = np.random.default_rng(5)
    x = rng.integers(0, 2, (3000, 16)).astype(np.float32)
    from vectordb_amd.builder import build_index
    fi = build_index(x, 6, 50, "l2", 5, seed=2)
    q = rng.integers(0, 2, (32, 16)).astype(np.float32)
    _check(fi, gpu, q, [(10, 3, 2), (5, 6, 7), (10, 6, 40)])
    # dim >= 255: u16 distances
    x = clustered(4000, 320, 30, 3, sigma=0.3)
    fi = build_index(x, 8, 40, "l2", 5, seed=3)
    _check(fi, gpu, queries_like(x, 24, 4, noise=0.1), [(10, 4, 3), (10, 8, 25)], worlds=(2,))


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_sharded_rabitq_equals_unsharded(gpu, metric):
    """RaBitQ codes: the shards exchange the keys of their own selections (estimate, leaf slot, place in the leaf);
    the result must be the unsharded search's bit for bit, also where the selection covers every candidate, where
    a shard holds none of a query's candidates, and for k = 100."""
    x, fi = small_index(n=6000, d=48, n_clusters=40, branching=8, leaf=40, metric=metric, code_kind="rabitq")
    q = queries_like(x, 48, 21)
    if metric != "cosine":
        q[7] = 0.0  # zero query: every estimate is norm^2 (rabitq.rs:408-413), ties decided by the scan position
    _check(fi, gpu, q, [(10, 4, 3), (10, 6, 10), (1, 2, 1), (20, 8, 1000), (100, 8, 2), (100, 5, 10)], worlds=(2, 3, 8))


def test_sharded_rabitq_duplicate_vectors(gpu):
    """Many identical vectors: equal estimates everywhere, so the leaf slot / in-leaf part of the keys decides."""
    rng = np.random.default_rng(9)
    base = rng.standard_normal((40, 32)).astype(np.float32)
    x = base[rng.integers(0, 40, 4000)]
    from vectordb_amd.builder import build_index
    fi = build_index(x, 6, 40, "l2", 5, seed=2, code_kind="rabitq")
    q = base[:24] + 0.01 * rng.standard_normal((24, 32)).astype(np.float32)
    _check(fi, gpu, q, [(10, 3, 2), (5, 6, 7), (10, 6, 40)])


def test_world_of_one_is_the_plain_search(gpu):
    x, fi = small_index(n=5000, d=32)
    dev = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(queries_like(x, 40, 2)).cuda(gpu)
    ids0, d0, c0 = dev.search_batch_device(dq, 10, 5, 4)
    ids0, d0, c0 = ids0.clone(), d0.clone(), c0.clone()
    ids1, d1, c1 = ExactShardedSearcher(dev).search_batch(dq, 10, 5, 4)
    torch.cuda.synchronize()
    assert torch.equal(ids0, ids1) and torch.equal(c0, c1)
    assert np.array_equal(_bits(d0.cpu().numpy()), _bits(d1.cpu().numpy()))
    x, fi = small_index(n=5000, d=32, code_kind="rabitq")
    dev = DeviceIndex(fi, gpu)
    ids0, d0, c0 = (t.clone() for t in dev.search_batch_device(dq, 10, 5, 4))
    ids1, d1, c1 = ExactShardedSearcher(dev).search_batch(dq, 10, 5, 4)
    torch.cuda.synchronize()
    assert torch.equal(ids0, ids1) and torch.equal(c0, c1)
    assert np.array_equal(_bits(d0.cpu().numpy()), _bits(d1.cpu().numpy()))


def test_route_device_matches_host_entry_point(gpu):
    from vectordb_amd import _ffi
    import ctypes as C

    x, fi = small_index(n=6000, d=48, n_clusters=40, branching=8, leaf=40)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 70, 3)
    dq = torch.from_numpy(q).cuda(gpu)
    for probes in (1, 4, 9, 1000):
        lv_h, lc_h = dev.route_batch(q, probes)
        lv_d, lc_d = dev.route_device(dq, probes)
        torch.cuda.synchronize()
        lv_d, lc_d = lv_d.cpu().numpy(), lc_d.cpu().numpy()
        assert lv_d.shape[1] == dev.leaf_stride(probes)
        assert np.array_equal(lc_h.astype(np.int32), lc_d)
        for i in range(q.shape[0]):
            assert np.array_equal(lv_h[i, :lc_h[i]].astype(np.int32), lv_d[i, :lc_d[i]]), (probes, i)
    # a leaf buffer narrower than the plan's lists is refused
    small = torch.zeros((70, 1), dtype=torch.int32, device=dq.device)
    cnt = torch.zeros(70, dtype=torch.int32, device=dq.device)
    st = dev._lib.cuddle_route_batch_device(dev._handle, C.c_void_p(dq.data_ptr()), 70, 48, 9, 0, C.c_void_p(small.data_ptr()), 1,
                                            C.c_void_p(cnt.data_ptr()), None)
    assert st == _ffi.ERR_BAD_ARG


@pytest.mark.parametrize("world,k,nq", [(2, 10, 33), (8, 10, 100), (3, 1, 5), (16, 64, 9), (5, 7, 257), (8, 100, 40), (11, 100, 4)])
def test_device_merge_matches_host_merge(gpu, world, k, nq):
    """cuddle_merge_topk_ordered_device against the host merge on lists with padding, equal distances
    (the order key decides) and signed zeros; shapes beyond the kernel's limits are refused."""
    import ctypes as C
    from vectordb_amd import _ffi

    rng = np.random.default_rng(world * 1000 + k)
    ids = rng.integers(0, 1 << 40, (world, nq, k)).astype(np.uint64)
    dist = np.round(rng.random((world, nq, k)) * 4).astype(np.float32) / 4  # many equal distances
    dist[rng.random(dist.shape) < 0.05] = -0.0
    dist.sort(axis=2)
    order = rng.permutation(world * nq * k).reshape(world, nq, k).astype(np.uint64)
    fill = rng.integers(0, k + 1, (world, nq))  # entries per list; the rest is padding
    pad = np.arange(k)[None, None, :] >= fill[:, :, None]
    ids[pad], dist[pad], order[pad] = np.uint64(0xFFFFFFFFFFFFFFFF), np.inf, np.uint64(0xFFFFFFFFFFFFFFFF)
    lib = _ffi.lib()
    h_ids = np.zeros((nq, k), dtype=np.uint64)
    h_dist = np.zeros((nq, k), dtype=np.float32)
    h_cnt = np.zeros(nq, dtype=np.uint32)
    _ffi.check(lib.cuddle_merge_topk_ordered(_ffi.ptr(ids), _ffi.ptr(dist), _ffi.ptr(order), world, nq, k,
                                             _ffi.ptr(h_ids), _ffi.ptr(h_dist), _ffi.ptr(h_cnt)))
    dev = torch.device("cuda", gpu)
    t_ids = torch.from_numpy(ids.view(np.int64)).to(dev)
    t_dist = torch.from_numpy(dist).to(dev)
    t_order = torch.from_numpy(order.view(np.int64)).to(dev)
    o_ids = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    st = lib.cuddle_merge_topk_ordered_device(t_ids.data_ptr(), t_dist.data_ptr(), t_order.data_ptr(), world, nq, k, 0,
                                              o_ids.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), None)
    if world * k > 1024:
        assert st == _ffi.ERR_UNSUPPORTED
        return
    _ffi.check(st)
    torch.cuda.synchronize()
    assert np.array_equal(o_cnt.cpu().numpy().astype(np.uint32), h_cnt)
    assert np.array_equal(o_ids.cpu().numpy().view(np.uint64), h_ids)
    # distances equal as floats (a signed zero may come from either of two tied lists only if their order keys tie, which they cannot)
    assert np.array_equal(_bits(o_dist.cpu().numpy()), _bits(h_dist))


def _many_small_leaves(n_leaves, per_leaf, dim, seed, code_kind="binary"):
    """A one-level index whose roots are its leaves: n_leaves leaves of per_leaf vectors each."""
    from vectordb_amd import _ffi
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    rng = np.random.default_rng(seed)
    cent = rng.uniform(-1, 1, (n_leaves, dim)).astype(np.float32)
    x = (np.repeat(cent, per_leaf, axis=0) + 0.1 * rng.standard_normal((n_leaves * per_leaf, dim))).astype(np.float32)
    thr = float(x.mean())
    return x, FlatIndex(dim=dim, metric=0, code_kind=_ffi.CODE_BINARY, centroids=cent,
                        child_offsets=np.zeros(n_leaves + 1, dtype=np.uint32), children=np.zeros(0, dtype=np.uint32),
                        roots=np.arange(n_leaves, dtype=np.uint32),
                        vec_offsets=(np.arange(n_leaves + 1, dtype=np.uint64) * per_leaf),
                        ids=np.arange(n_leaves * per_leaf, dtype=np.uint64), vectors=x, threshold=thr,
                        codes=pack_binary_codes(x, thr))


def _two_levels_of_small_leaves(n_top, fan, per_leaf, dim, seed):
    """n_top roots with `fan` leaf children each, per_leaf vectors per leaf (node ids: roots first, then the leaves)."""
    from vectordb_amd import _ffi
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    rng = np.random.default_rng(seed)
    top = rng.uniform(-1, 1, (n_top, dim)).astype(np.float32)
    leaf_c = (np.repeat(top, fan, axis=0) + 0.3 * rng.standard_normal((n_top * fan, dim))).astype(np.float32)
    n_leaves = n_top * fan
    x = (np.repeat(leaf_c, per_leaf, axis=0) + 0.05 * rng.standard_normal((n_leaves * per_leaf, dim))).astype(np.float32)
    thr = float(x.mean())
    child_offsets = np.concatenate([np.arange(n_top + 1, dtype=np.uint32) * fan, np.full(n_leaves, n_leaves, dtype=np.uint32)])
    vec_offsets = np.concatenate([np.zeros(n_top, dtype=np.uint64), np.arange(n_leaves + 1, dtype=np.uint64) * per_leaf])
    return x, FlatIndex(dim=dim, metric=0, code_kind=_ffi.CODE_BINARY, centroids=np.concatenate([top, leaf_c]),
                        child_offsets=child_offsets, children=(n_top + np.arange(n_leaves)).astype(np.uint32),
                        roots=np.arange(n_top, dtype=np.uint32), vec_offsets=vec_offsets,
                        ids=np.arange(n_leaves * per_leaf, dtype=np.uint64), vectors=x, threshold=thr,
                        codes=pack_binary_codes(x, thr))


def test_sharded_leaf_lists_beyond_64k_of_lds(gpu):
    """Thousands of reached leaves per query: the leaf tables of the selection kernels (16 bytes per leaf slot plus
    the query) pass the 64 KiB a kernel gets without asking for more, and the routing's last step has to fit its
    candidates, expansion scratch and segment table into the CU's LDS or leave the level to the exact kernels.
    Unsharded against the oracle, sharded against unsharded."""
    from test_gpu_parity import _compare_full
    x, fi = _many_small_leaves(5000, 2, 48, 3)
    q = queries_like(x, 9, 5)
    _compare_full(fi, DeviceIndex(fi, gpu), q, 10, 4096, 4)   # one level of 5000 > probes + margin: exact kernels
    _check(fi, gpu, q, [(10, 4096, 4)], worlds=(2,))
    x, fi = _two_levels_of_small_leaves(64, 100, 2, 32, 4)
    q = queries_like(x, 7, 6)
    _compare_full(fi, DeviceIndex(fi, gpu), q, 10, 3500, 2)   # shared level of 6400 through the filter, 3940 candidates
    _compare_full(fi, DeviceIndex(fi, gpu), q, 10, 40, 2)     # per-query frontiers of 4000 through the grouped filter
    _check(fi, gpu, q, [(10, 3500, 2)], worlds=(3,))


def test_sharded_batch_larger_than_one_round(gpu):
    """More than 4096 queries: ExactShardedSearcher cuts the batch into equal rounds."""
    x, fi = small_index(n=3000, d=24, n_clusters=15, branching=6, leaf=25)
    q = np.tile(queries_like(x, 100, 4), (45, 1))[:4300]
    full = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(q).cuda(gpu)
    ids0, d0, c0 = [t.clone() for t in full.search_batch_device(dq, 10, 4, 3)]
    ids1, d1, c1 = ExactShardedSearcher(full).search_batch(dq, 10, 4, 3)  # a world of one, two rounds
    torch.cuda.synchronize()
    assert ids1.shape[0] == 4300 and torch.equal(ids0, ids1) and torch.equal(c0, c1)
    assert np.array_equal(_bits(d0.cpu().numpy()), _bits(d1.cpu().numpy()))


def test_sharded_rabitq_many_leaves_and_a_selection_near_the_cap(gpu):
    """probes >= 1000 and a selection close to 10 000: every member's leaf slot (part of its exchange key) is found
    among well over a thousand leaf slots; the keys must still order candidates as the unsharded selection does."""
    rng = np.random.default_rng(11)
    n, d, nl = 48000, 32, 1600
    x = rng.standard_normal((n, d)).astype(np.float32)
    from vectordb_amd import _ffi
    from vectordb_amd.builder import make_rotation, rabitq_encode
    from vectordb_amd.flat_index import FlatIndex
    rot = make_rotation(d, 5)
    codes, norms, ipoo = rabitq_encode(torch.from_numpy(x), torch.from_numpy(rot))
    cent = x.reshape(nl, n // nl, d).mean(axis=1)
    fi = FlatIndex(dim=d, metric=_ffi.METRIC_L2, code_kind=_ffi.CODE_RABITQ, centroids=cent, child_offsets=np.zeros(nl + 1, np.uint32),
                   children=np.zeros(0, np.uint32), roots=np.arange(nl, dtype=np.uint32),
                   vec_offsets=(np.arange(nl + 1) * (n // nl)).astype(np.uint64), ids=np.arange(n, dtype=np.uint64), vectors=x,
                   rabitq_rot_t=rot, rabitq_codes=np.asarray(codes), rabitq_norm=np.asarray(norms), rabitq_ip_oo=np.asarray(ipoo))
    q = x[rng.integers(0, n, 12)] + 0.05 * rng.standard_normal((12, d)).astype(np.float32)
    _check(fi, gpu, q, [(100, 1200, 95)], worlds=(3,))
