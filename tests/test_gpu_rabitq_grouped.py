"""GPU: the RaBitQ bounds computed leaf by leaf on the matrix cores (CUDDLE_RABITQ_GROUPED, off by default) against
the per-query kernel's. Mode 2 recomputes every bound in the per-query kernel and fails the search if one differs in
a single bit; mode 1 must give the search results of the default path and of the oracle
(reference: src/index/hierarchical_with_rabitq.rs:329-459, src/quantization/rabitq.rs:335-363)."""
import numpy as np
import pytest

from vectordb_amd.index import DeviceIndex

from helpers import queries_like, small_index

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,metric", [(64, "l2"), (100, "cosine"), (40, "dot")])
def test_grouped_bounds_equal_the_per_query_bounds(gpu, monkeypatch, d, metric):
    from oracle import oracle as orc
    x, fi = small_index(n=9000, d=d, n_clusters=50, branching=8, leaf=45, metric=metric, code_kind="rabitq")
    q = queries_like(x, 150, 5)
    if metric != "cosine":  # (a zero query under cosine is a NaN distance: the reference panics there)
        q[7] = 0.0  # a zero query takes no bounds at all
    dev = DeviceIndex(fi, gpu)
    ref = dev.search_batch(q, 10, 6, 4)           # default path: bounds inside the per-query kernel
    for mode in ("2", "1"):
        monkeypatch.setenv("CUDDLE_RABITQ_GROUPED", mode)
        got = dev.search_batch(q, 10, 6, 4)       # mode 2 raises if any bound differs
        for a, b in zip(ref, got):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    monkeypatch.delenv("CUDDLE_RABITQ_GROUPED")
    oi = orc.OracleIndex(fi)
    rc, ids, dd, cnt = oi.search_batch(q, 10, 6, 4)
    assert rc == 0 and np.array_equal(cnt, ref[2])
    mask = np.arange(10)[None, :] < cnt[:, None]
    assert np.array_equal(ids[mask], ref[0][mask]) and np.array_equal(dd.view(np.uint32)[mask], ref[1].view(np.uint32)[mask])
