"""GPU: the RCCL ("nccl") backend at world size 1 — process-group initialisation on the device and
every collective the sharded search and the sharded index generator use (all-reduce, all-gather,
all-gather-into-tensor, broadcast, barrier), then the exact sharded searcher on top of it. A one-GPU
box cannot show more ranks; the N > 1 protocol itself is covered by the gloo tests and by the
in-process shard tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from vectordb_amd.index import DeviceIndex
from vectordb_amd.sharded import ExactShardedSearcher, _all_gather, _all_reduce_sum

from helpers import queries_like, small_index

pytestmark = pytest.mark.gpu


def test_rccl_collectives_and_sharded_search_world1(gpu):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    device = torch.device("cuda", gpu)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        assert dist.get_backend() == "nccl"
        hist = torch.arange(4096 * 129, dtype=torch.int32, device=device).reshape(4096, 129)
        ref = hist.clone()
        _all_reduce_sum(hist, None)
        mine = torch.arange(1000 * 10, dtype=torch.int64, device=device).reshape(1000, 10)
        out = torch.empty((1, 1000, 10), dtype=torch.int64, device=device)
        _all_gather(out, mine, None)
        flat = torch.empty((1000, 10), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(flat, mine)
        b = torch.full((7,), 3.0, device=device)
        dist.broadcast(b, src=0)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(hist, ref) and torch.equal(out[0], mine) and torch.equal(flat, mine) and float(b.sum()) == 21.0
        # the searcher over the group: a world of one is the plain search
        x, fi = small_index(n=5000, d=32)
        dev = DeviceIndex(fi, gpu)
        dq = torch.from_numpy(queries_like(x, 40, 2)).to(device)
        ids0, d0, c0 = [t.clone() for t in dev.search_batch_device(dq, 10, 5, 4)]
        searcher = ExactShardedSearcher(dev)
        assert searcher.world == 1 and searcher.rank == 0
        ids1, d1, c1 = searcher.search_batch(dq, 10, 5, 4)
        torch.cuda.synchronize()
        assert torch.equal(ids0, ids1) and torch.equal(c0, c1)
        assert np.array_equal(d0.cpu().numpy().view(np.uint32), d1.cpu().numpy().view(np.uint32))
        # the sharded generator's collectives (world 1: none are needed, the path must not call them)
        from vectordb_amd.synthetic import build_tree_index_on_device
        dti, q = build_tree_index_on_device(32, 6, 2, 10, device, seed=1, n_queries=8, rank=0, world=1)
        assert dti.n_vectors == 360 and q.shape == (8, 32)
    finally:
        dist.destroy_process_group()


def test_c_abi_communicator_world1(gpu):
    """cuddle_comm_* / cuddle_sharded_search_batch: RCCL loaded and called by the library itself. A
    communicator of one rank runs all four collectives of the protocol and must return the plain search."""
    from vectordb_amd import _ffi
    from vectordb_amd.sharded import RcclShardedSearcher

    device = torch.device("cuda", gpu)
    uid = RcclShardedSearcher.make_unique_id()
    assert len(uid) == 128 and any(uid)
    for metric, d, kind in (("l2", 48, "binary"), ("cosine", 320, "binary"), ("l2", 64, "rabitq")):
        x, fi = small_index(n=6000, d=d, n_clusters=40, branching=8, leaf=40, metric=metric, code_kind=kind)
        dev = DeviceIndex(fi, gpu)
        dq = torch.from_numpy(queries_like(x, 70, 3)).to(device)
        s = RcclShardedSearcher(dev, 0, 1, uid if d == 48 else RcclShardedSearcher.make_unique_id())
        for (k, probes, rf) in ((10, 4, 3), (1, 2, 1), (20, 8, 1000), (100, 6, 10)):
            ids0, d0, c0 = [t.clone() for t in dev.search_batch_device(dq, k, probes, rf)]
            ids1, d1, c1 = s.search_batch(dq, k, probes, rf)
            torch.cuda.synchronize()
            assert torch.equal(ids0, ids1) and torch.equal(c0, c1), (metric, k, probes, rf)
            assert np.array_equal(d0.cpu().numpy().view(np.uint32), d1.cpu().numpy().view(np.uint32))
        if d == 48:  # more than 4096 queries: equal rounds of the protocol, one after the other
            big = dq.repeat(65, 1)[:4500].contiguous()
            ids0, d0, c0 = [t.clone() for t in dev.search_batch_device(big, 10, 4, 3)]
            ids1, d1, c1 = s.search_batch(big, 10, 4, 3)
            torch.cuda.synchronize()
            assert torch.equal(ids0, ids1) and torch.equal(c0, c1)
            assert np.array_equal(d0.cpu().numpy().view(np.uint32), d1.cpu().numpy().view(np.uint32))
        # limits of the merge kernel are refused, not mis-computed
        with pytest.raises(_ffi.CuddleError) as e:
            s.search_batch(dq, 1025, 4, 3)
        assert e.value.status == _ffi.ERR_UNSUPPORTED
        s.close()
