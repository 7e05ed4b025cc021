"""GPU: cuddle_search_batch_device_noalloc (work space in the library's fixed per-GPU array, no
allocation) must give cuddle_search_batch_device's results bit for bit — on two calls in a row,
the second with a smaller input, and for both code families; and that it allocates nothing: unchanged device memory on a
handle that never ran the allocating call, two streams in turn, and a run under stream capture."""
import ctypes as C

import numpy as np
import pytest
import torch

from vectordb_amd import _ffi
from vectordb_amd.index import DeviceIndex

from helpers import queries_like, small_index

pytestmark = pytest.mark.gpu


def _bits(t):
    return np.ascontiguousarray(t.cpu().numpy(), dtype=np.float32).view(np.uint32)


def _both(dev, dq, k, probes, rf, flags=0):
    nq = dq.shape[0]
    ref = dev.search_batch_device(dq, k, probes, rf, flags)
    ref = [t.clone() for t in ref]
    out = (torch.full((nq, k), 7, dtype=torch.int64, device=dq.device), torch.full((nq, k), -1.0, device=dq.device),
           torch.full((nq,), 99, dtype=torch.int32, device=dq.device))
    got = dev.search_batch_device_noalloc(dq, k, probes, rf, flags, *out)
    torch.cuda.synchronize()
    assert torch.equal(ref[2].cpu(), got[2].cpu())
    assert torch.equal(ref[0].cpu(), got[0].cpu())
    assert np.array_equal(_bits(ref[1]), _bits(got[1]))


@pytest.mark.parametrize("code_kind,metric", [("binary", "l2"), ("rabitq", "cosine")])
def test_noalloc_matches_search_batch_device(gpu, code_kind, metric):
    x, fi = small_index(n=12000, d=64, n_clusters=60, branching=10, leaf=40, metric=metric, code_kind=code_kind)
    dev = DeviceIndex(fi, gpu)
    q = queries_like(x, 300, 3)
    dq = torch.from_numpy(q).cuda(gpu)
    _both(dev, dq, 10, 8, 20)            # first call
    _both(dev, dq[:37].contiguous(), 5, 3, 4)   # second call: smaller batch, smaller k / probes / selection
    _both(dev, dq, 10, 8, 20, _ffi.SEARCH_STATS_VARIANT)
    # a second handle on the same GPU shares the array: calls one after the other stay correct
    x2, fi2 = small_index(n=3000, d=32, n_clusters=20, branching=6, leaf=30, seed=4)
    dev2 = DeviceIndex(fi2, gpu)
    dq2 = torch.from_numpy(queries_like(x2, 50, 1)).cuda(gpu)
    _both(dev2, dq2, 10, 4, 5)
    _both(dev, dq[:100].contiguous(), 10, 8, 20)


def test_noalloc_chunks_what_does_not_fit_at_once(gpu):
    # 300 000 candidates per query: the distances spill to the work space, and a chunk of all 64 queries
    # would need more than the array holds, so the batch is cut into smaller chunks
    rng = np.random.default_rng(0)
    n, d = 300000, 16
    x = rng.standard_normal((n, d)).astype(np.float32)
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    # 150 leaves of 2000 vectors under one root level: probes 150 -> 300 000 candidates per query
    nl = 150
    cent = x.reshape(nl, n // nl, d).mean(axis=1)
    fi = FlatIndex(dim=d, metric=_ffi.METRIC_L2, code_kind=_ffi.CODE_BINARY, centroids=cent,
                   child_offsets=np.zeros(nl + 1, np.uint32), children=np.zeros(0, np.uint32), roots=np.arange(nl, dtype=np.uint32),
                   vec_offsets=(np.arange(nl + 1) * (n // nl)).astype(np.uint64), ids=np.arange(n, dtype=np.uint64), vectors=x,
                   threshold=0.0, codes=pack_binary_codes(x, 0.0))
    dev = DeviceIndex(fi, gpu)
    dq = torch.from_numpy(x[:64].copy()).cuda(gpu)
    _both(dev, dq, 10, 150, 100)  # 64 queries x 300 000 one-byte distances do not fit 32 MiB at once: chunked, still equal
    lib = _ffi.lib()
    out = (torch.zeros((64, 10), dtype=torch.int64, device=dq.device), torch.zeros((64, 10), device=dq.device),
           torch.zeros((64,), dtype=torch.int32, device=dq.device))
    st = lib.cuddle_search_batch_device_noalloc(dev._handle, C.c_void_p(dq.data_ptr()), 64, d, 10, 150, 100, 0,
                                                C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                                C.c_void_p(out[2].data_ptr()), None)
    assert st == _ffi.OK
    st = lib.cuddle_search_batch_device_noalloc(dev._handle, C.c_void_p(dq.data_ptr()), 64, d + 1, 10, 150, 100, 0,
                                                C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                                C.c_void_p(out[2].data_ptr()), None)
    assert st == _ffi.ERR_BAD_DIM


def test_noalloc_really_allocates_nothing(gpu):
    """A fresh handle that has never run the allocating entry point: the no-allocation call must leave the handle's
    device bytes and the GPU's free memory as they were, and must run inside a stream capture in global mode, where
    any hipMalloc / hipFree / host wait fails the capture. The captured graph, replayed, gives the same results."""
    x, fi = small_index(n=12000, d=64, n_clusters=60, branching=10, leaf=40, metric="l2", code_kind="binary")
    dq = torch.from_numpy(queries_like(x, 300, 3)).cuda(gpu)
    ref_dev = DeviceIndex(fi, gpu)
    ref = [t.clone() for t in ref_dev.search_batch_device(dq, 10, 8, 20)]
    torch.cuda.synchronize()
    ref_dev.close()

    dev = DeviceIndex(fi, gpu)  # never sees search_batch_device: its work space stays unallocated
    out = (torch.full((300, 10), 7, dtype=torch.int64, device=dq.device), torch.full((300, 10), -1.0, device=dq.device),
           torch.full((300,), 99, dtype=torch.int32, device=dq.device))
    torch.cuda.synchronize()
    bytes_before = dev.device_bytes
    free_before = torch.cuda.mem_get_info(gpu)[0]
    dev.search_batch_device_noalloc(dq, 10, 8, 20, 0, *out)
    torch.cuda.synchronize()
    assert dev.device_bytes == bytes_before
    assert torch.cuda.mem_get_info(gpu)[0] == free_before
    assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2]) and np.array_equal(_bits(out[1]), _bits(ref[1]))

    # two streams in turn: the second call is ordered behind the first on the device
    s1, s2 = torch.cuda.Stream(gpu), torch.cuda.Stream(gpu)
    out2 = tuple(torch.zeros_like(t) for t in out)
    s1.wait_stream(torch.cuda.current_stream(gpu))
    s2.wait_stream(torch.cuda.current_stream(gpu))
    dev.search_batch_device_noalloc(dq, 10, 8, 20, 0, *out, stream=s1.cuda_stream)
    dev.search_batch_device_noalloc(dq[:100].contiguous(), 10, 8, 20, 0, out2[0][:100], out2[1][:100], out2[2][:100],
                                    stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and np.array_equal(_bits(out[1]), _bits(ref[1]))
    assert torch.equal(out2[0][:100], ref[0][:100]) and np.array_equal(_bits(out2[1][:100]), _bits(ref[1][:100]))

    # under capture: an allocation, a free or a host-side wait inside the call would end the capture with an error
    for t in out:
        t.fill_(0)
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(gpu)
    cs.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.graph(g, stream=cs, capture_error_mode="global"):
        dev.search_batch_device_noalloc(dq, 10, 8, 20, 0, *out, stream=torch.cuda.current_stream(gpu).cuda_stream)
    torch.cuda.synchronize()
    assert int(out[2].sum()) == 0  # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2]) and np.array_equal(_bits(out[1]), _bits(ref[1]))
    for t in out:
        t.fill_(0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and np.array_equal(_bits(out[1]), _bits(ref[1]))
    assert dev.device_bytes == bytes_before
