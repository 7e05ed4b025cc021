"""GPU: the RaBitQ bound pass cuts a query's candidates into tiles of 64 scan positions, whatever leaves they fall
into (a tile inside one leaf takes the wave-uniform path, a tile across leaf ends gives every lane its own leaf).
The results must be those of tiles cut at leaf ends (CUDDLE_RQ_PACK=0) and of the oracle: leaves of one to a few
hundred vectors, empty leaves, more tiles than the per-tile slot table holds
(reference: compute_rabitq_distances / select_top_candidates, src/index/hierarchical_with_rabitq.rs:581-633)."""
import os
import sys

import numpy as np
import pytest

from vectordb_amd.index import DeviceIndex

from helpers import queries_like, small_index

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _same(a, b):
    return all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
               for x, y in zip(a, b))


def _check(gpu, monkeypatch, fi, q, k, probes, rf):
    from oracle import oracle as orc
    dev = DeviceIndex(fi, gpu)
    packed = dev.search_batch(q, k, probes, rf)
    monkeypatch.setenv("CUDDLE_RQ_PACK", "0")
    at_leaf_ends = dev.search_batch(q, k, probes, rf)
    monkeypatch.delenv("CUDDLE_RQ_PACK")
    assert _same(packed, at_leaf_ends)
    rc, ids, dd, cnt = orc.OracleIndex(fi).search_batch(q, k, probes, rf)
    assert rc == 0 and np.array_equal(cnt, packed[2])
    mask = np.arange(k)[None, :] < cnt[:, None]
    assert np.array_equal(ids[mask], packed[0][mask]) and np.array_equal(dd.view(np.uint32)[mask], packed[1].view(np.uint32)[mask])


@pytest.mark.parametrize("d,metric,leaf", [(64, "l2", 45), (100, "cosine", 103), (40, "dot", 10), (768, "cosine", 150)])
def test_tiles_across_leaf_ends(gpu, monkeypatch, d, metric, leaf):
    x, fi = small_index(n=9000, d=d, n_clusters=50, branching=8, leaf=leaf, metric=metric, code_kind="rabitq")
    _check(gpu, monkeypatch, fi, queries_like(x, 120, 5), 10, 12, 4)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tiles_over_an_irregular_forest(gpu, monkeypatch, seed):
    """Leaves of 0, 1, 2, 3, 63, 64, 65, 200 and 1500 vectors at any depth: tiles that hold dozens of leaves, tiles
    that hold a piece of one."""
    from random_parity_sweep import random_tree_index
    rng = np.random.default_rng(seed)
    x, fi = random_tree_index(rng, 48, "l2", "rabitq")
    if x.shape[0] < 50:
        pytest.skip("the drawn forest is nearly empty")
    _check(gpu, monkeypatch, fi, queries_like(x, 64, seed + 10), 10, 40, 3)


def test_more_tiles_than_the_slot_table_holds(gpu, monkeypatch):
    """Over 262 144 candidates per query: the slot of a tile's first position is searched for instead of looked up,
    and the bounds live in the spill rows."""
    x, fi = small_index(n=330_000, d=32, n_clusters=40, branching=4, leaf=5000, metric="l2", code_kind="rabitq")
    q = queries_like(x, 6, 9)
    dev = DeviceIndex(fi, gpu)
    _, _, _, st = dev.search_batch(q, 10, 4096, 20, want_stats=True)
    assert st["max_candidates"] > 262_144, st
    _check(gpu, monkeypatch, fi, q, 10, 4096, 20)
