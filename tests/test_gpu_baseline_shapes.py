"""GPU parity at the SHAPES of the benchmark configurations (reduced n, right dimension, metric, k,
probes, rerank factors and code family), against the oracle:

  * 960-d, L2, k = 10, probes 200, rerank 5/10/100   (GIST1M; /root/reference/benches/gist_benchmark.rs:177-183)
    - the 15-word Hamming scan instantiation, the 960-d rerank and the 960-d matrix-core routing level;
  * 768-d, cosine, RaBitQ codes, k = 100, probes 100, rerank 10   (/root/reference/src/python_bindings.rs:178)
    - on an index generated and encoded on the device (the generator of the 10M benchmark run);
  * 128-d, L2, k = 10, a device-built three-level tree split over 8 leaf shards (the 100M / 1B configs)
    - u8 distances, grouped routing, shards with empty leaves, exact sharded protocol.
The committed fixtures of the same shapes (tests/golden/*_shape_* , rabitq_cosine_d768_k100) are
covered by test_golden.py; the 8-way split is checked against the fixture here as well.
"""
import numpy as np
import pytest
import torch

import golden_util
from vectordb_amd import _ffi
from vectordb_amd.builder import build_index
from vectordb_amd.index import DeviceIndex
from vectordb_amd.sharded import exact_search_in_process
from vectordb_amd.synthetic import build_tree_index_on_device

from helpers import clustered, queries_like
from test_gpu_parity import _compare_full, _oracle

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_gist_shape_960d_binary(gpu):
    x = clustered(30000, 960, 300, 21, sigma=0.3)
    fi = build_index(x, 100, 30, "l2", 6, seed=2)
    widths, cur = [], list(fi.roots)
    while cur:
        widths.append(len(cur))
        cur = [c for v in cur for c in fi.children[fi.child_offsets[v]:fi.child_offsets[v + 1]]]
    assert widths[1] >= 512 and fi.codes.shape[1] == 15, widths  # a level the matrix cores route; 15-word codes
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 48, 5, noise=0.1)
    for rf in (5, 10, 100):
        _compare_full(fi, dev, q, 10, 200, rf)
    # stage outputs at the benchmark's operating point
    orc = _oracle()
    oi = orc.OracleIndex(fi)
    leaves, lcnt = dev.route_batch(q, 200)
    pos, key, ccnt = dev.scan_batch(q, leaves, lcnt)
    _, _, _, spos, skey, sdist, scnt = dev.search_leaves_batch(q, leaves, lcnt, 10, 10)
    for i in range(0, q.shape[0], 5):
        t = oi.search(q[i], 10, 200, 10, trace=True)
        assert np.array_equal(leaves[i, :lcnt[i]], t["leaves"]), i
        assert np.array_equal(pos[i, :ccnt[i]], t["cand_pos"]) and np.array_equal(key[i, :ccnt[i]], t["cand_key"]), i
        assert np.array_equal(spos[i, :scnt[i]], t["sel_pos"]) and np.array_equal(skey[i, :scnt[i]], t["sel_key"]), i
        assert np.array_equal(_bits(sdist[i, :scnt[i]]), _bits(t["sel_dist"])), i
    st = dev.search_batch(q, 10, 200, 10, want_stats=True)[3]
    assert st["max_candidates"] > 200 * 10  # the selection is a real cut, not "keep everything"


def test_rabitq_shape_768d_cosine_k100_device_encoded(gpu):
    dev_t = torch.device("cuda", gpu)
    dti, dq = build_tree_index_on_device(768, 10, 3, 30, dev_t, seed=11, metric=_ffi.METRIC_COSINE, style="gauss",
                                         n_queries=40, code_kind=_ffi.CODE_RABITQ)
    dev = dti.create(gpu)
    fi = dti.to_flat()
    assert fi.n_vectors == 30000 and fi.rabitq_codes.shape[1] == 24
    q = dq.cpu().numpy()
    _compare_full(fi, dev, q, 100, 100, 10)
    # the host-uploaded copy of the same index gives the same answers as the device-adopted one
    dev2 = DeviceIndex(fi, gpu)
    a, b = dev.search_batch(q, 100, 100, 10), dev2.search_batch(q, 100, 100, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1])) and np.array_equal(a[2], b[2])
    orc = _oracle()
    oi = orc.OracleIndex(fi)
    leaves, lcnt = dev.route_batch(q, 100)
    pos, key, ccnt = dev2.scan_batch(q, leaves, lcnt)
    for i in range(0, q.shape[0], 8):
        t = oi.search(q[i], 100, 100, 10, trace=True)
        assert np.array_equal(leaves[i, :lcnt[i]], t["leaves"]), i
        assert np.array_equal(pos[i, :ccnt[i]], t["cand_pos"]) and np.array_equal(key[i, :ccnt[i]], t["cand_key"]), i


def test_rabitq_selection_of_10000(gpu):
    """k * rerank_factor = 10000 with more than 10000 candidates per query: the largest phase-1
    selection `search` allows (hierarchical_with_rabitq.rs:443) through the f32 selection sort."""
    dev_t = torch.device("cuda", gpu)
    dti, dq = build_tree_index_on_device(64, 12, 2, 100, dev_t, seed=5, metric=_ffi.METRIC_L2, style="gauss",
                                         n_queries=6, code_kind=_ffi.CODE_RABITQ)
    dev = dti.create(gpu)
    fi = dti.to_flat()
    q = dq.cpu().numpy()
    ids, dist, cnt, st = dev.search_batch(q, 100, 144, 100, want_stats=True)
    assert st["max_candidates"] == 14400 and st["vectors_reranked"] == 6 * 10000
    _compare_full(fi, dev, q, 100, 144, 100)
    _compare_full(fi, dev, q[:2], 100, 144, 160, _ffi.SEARCH_STATS_VARIANT)  # uncapped: 14400 selected, 16384-slot sort


def _sharded_equals(fi, gpu, q, k, probes, rf, world):
    full = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(q).cuda(gpu)
    shards = [DeviceIndex(fi.shard(r, world), gpu) for r in range(world)]
    ids0, d0, c0 = full.search_batch_device(dq, k, probes, rf)
    for shard_routing, packed in ((False, False), (True, True)):
        ids1, d1, c1 = exact_search_in_process(shards, dq, k, probes, rf, shard_routing=shard_routing, packed=packed)
        torch.cuda.synchronize()
        assert torch.equal(c0.cpu(), c1.cpu()) and torch.equal(ids0.cpu(), ids1.cpu()), (shard_routing, packed)
        assert np.array_equal(_bits(d0.cpu().numpy()), _bits(d1.cpu().numpy()))
    return ids0.cpu().numpy().view(np.uint64), d0.cpu().numpy(), c0.cpu().numpy()


def test_shard_shape_128d_8way_device_built_tree(gpu):
    dev_t = torch.device("cuda", gpu)
    dti, dq = build_tree_index_on_device(128, 20, 3, 12, dev_t, seed=3, metric=_ffi.METRIC_L2, style="sift", n_queries=64)
    fi = dti.to_flat()
    assert fi.n_vectors == 96000 and fi.n_levels() == 3
    q = dq.cpu().numpy()
    for (k, probes, rf) in ((10, 100, 25), (10, 100, 5), (10, 30, 50)):
        ids, dist, cnt = _sharded_equals(fi, gpu, q, k, probes, rf, 8)
        oi = _oracle().OracleIndex(fi)
        for i in range(0, q.shape[0], 4):
            r = oi.search(q[i], k, probes, rf)
            assert r["count"] == cnt[i] and np.array_equal(ids[i, :cnt[i]], r["ids"][:cnt[i]]), (k, probes, rf, i)
            assert np.array_equal(_bits(dist[i, :cnt[i]]), _bits(r["dist"][:cnt[i]]))


def test_shard_shape_golden_8way(gpu):
    fi, q, par, exp = golden_util.load("shard_shape_l2_d128")
    ids, dist, cnt = _sharded_equals(fi, gpu, q, par["k"], par["probes"], par["rf"], 8)
    assert np.array_equal(cnt.astype(np.uint32), exp["count"])
    assert np.array_equal(ids, exp["ids"]) and np.array_equal(_bits(dist), _bits(exp["dist"]))
