"""Index-build side on the GPU: cuddle_kmeans_assign against the oracle's KMeans::assign
(reference: src/clustering/kmeans.rs:136-160). Bit-exact: assignment and distance."""
import numpy as np
import pytest
import torch

from vectordb_amd import _ffi
from vectordb_amd.builder import kmeans_assign, metric_id

from helpers import clustered

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


def _check(x, c, metric, max_fallbacks=None):
    orc = _oracle()
    dev = torch.device("cuda:0")
    a, d, fb = kmeans_assign(torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev), metric)
    ea, ed = orc.kmeans_assign(x, c, metric_id(metric))
    assert np.array_equal(a.cpu().numpy().astype(np.uint32), ea)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    if max_fallbacks is not None:
        assert fb <= max_fallbacks, fb
    return fb


def test_kmeans_kats(gpu):  # kmeans.rs:256-310
    cent = np.array([[0.5, 0.5], [10.0, 10.0]], dtype=np.float32)
    pts = np.array([[1.0, 0.0], [0.0, 1.0], [10.0, 10.0], [0.5, 0.5]], dtype=np.float32)
    dev = torch.device("cuda:0")
    a, d, _ = kmeans_assign(torch.from_numpy(pts).to(dev), torch.from_numpy(cent).to(dev), "l2")
    assert a.tolist() == [0, 0, 1, 0] and d.tolist() == [0.5, 0.5, 0.0, 0.0]


@pytest.mark.parametrize("metric,n,k,d", [("l2", 20000, 100, 128), ("l2", 5000, 1000, 37), ("cosine", 9000, 300, 96),
                                          ("dot", 7000, 257, 64), ("l2", 3000, 70, 960), ("cosine", 130, 3, 5),
                                          ("l2", 1, 1, 1), ("dot", 4097, 129, 200)])
def test_kmeans_assign_matches_oracle(gpu, metric, n, k, d):
    x = clustered(n, d, max(2, min(k, 50)), 7, sigma=0.3)
    rng = np.random.default_rng(3)
    c = x[rng.integers(0, n, k)] + 0.05 * rng.standard_normal((k, d)).astype(np.float32)
    _check(x, c.astype(np.float32), metric, max_fallbacks=n // 20 + 1)


def test_kmeans_assign_ties_and_duplicates(gpu):
    """Duplicate centroids and points that are centroids: equal minima go to the lowest index. Duplicates
    in different segments, or two in one, are separate candidates; three in one 64-centroid tile (20, 21,
    22) force the exact comparison with the whole segment."""
    rng = np.random.default_rng(5)
    c = rng.standard_normal((200, 32)).astype(np.float32)
    c[150] = c[20]
    c[77] = c[20]
    c[21] = c[20]
    c[22] = c[20]
    c[199] = c[0]
    x = np.concatenate([c[[20, 0, 150, 5]], rng.standard_normal((500, 32)).astype(np.float32)])
    fb = _check(x, c, "l2")
    assert fb >= 2  # the points equal to centroids 20 / 150
    grid = (rng.integers(0, 3, (400, 8))).astype(np.float32)  # many exactly equal distances
    _check(grid, grid[:64].copy(), "l2")
    _check(grid + 1.0, grid[:64] + 1.0, "dot")


def test_kmeans_assign_far_from_origin_and_scaled(gpu):
    x = clustered(6000, 64, 40, 9, sigma=0.3)
    rng = np.random.default_rng(1)
    c = x[rng.integers(0, 6000, 500)].copy()
    _check(x + 100.0, c + 100.0, "l2", max_fallbacks=6000)
    _check(x * 1000.0, c * 1000.0, "l2", max_fallbacks=6000)
    _check((x + 3.0) * 0.01, (c + 3.0) * 0.01, "cosine", max_fallbacks=6000)


def test_kmeans_assign_large(gpu):
    """More than one pass of 262144 vectors; checked on a sample (the oracle is O(n k d))."""
    orc = _oracle()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn((300000, 64), generator=g, device=dev)
    c = x[torch.randint(0, 300000, (2000,), generator=g, device=dev)] + 0.1 * torch.randn((2000, 64), generator=g, device=dev)
    a, d, fb = kmeans_assign(x, c, "l2")
    sel = torch.cat([torch.arange(0, 300000, 997, device=dev), torch.arange(262100, 262200, device=dev), torch.tensor([299999], device=dev)])
    ea, ed = orc.kmeans_assign(x[sel].cpu().numpy(), c.cpu().numpy(), _ffi.METRIC_L2)
    assert np.array_equal(a[sel].cpu().numpy().astype(np.uint32), ea)
    assert np.array_equal(d[sel].cpu().numpy().view(np.uint32), ed.view(np.uint32))
    assert fb < 3000


def test_kmeans_assign_errors(gpu):
    dev = torch.device("cuda:0")
    x = torch.zeros((4, 8), device=dev)
    with pytest.raises(ValueError):
        kmeans_assign(x, torch.zeros((2, 7), device=dev), "l2")
    lib = _ffi.lib()
    assert lib.cuddle_kmeans_assign(None, 4, x.data_ptr(), 2, 8, 0, x.data_ptr(), x.data_ptr(), None, 0, None) == _ffi.ERR_BAD_ARG
    assert lib.cuddle_kmeans_assign(x.data_ptr(), 4, x.data_ptr(), 0, 8, 0, x.data_ptr(), x.data_ptr(), None, 0, None) == _ffi.ERR_BAD_ARG
    assert lib.cuddle_kmeans_assign(x.data_ptr(), 4, x.data_ptr(), 2, 8, 9, x.data_ptr(), x.data_ptr(), None, 0, None) == _ffi.ERR_BAD_ARG
