"""Leaf-sharded search over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm).

Partitioning: every rank holds the whole centroid tree (it is small) and the codes + fp32 vectors
of the leaves j with j % world == rank (`FlatIndex.shard`). Every rank routes the full batch
identically, scans and reranks only its own leaves, and returns its local top-k. The one exchange
step of the path is a single all-gather of the [n_queries, k] (distance, id) lists per batch —
a few hundred KB, latency-bound on the fully connected xGMI links — followed by a k-way merge
that every rank performs identically (ties: distance, then rank, then position in the rank's
list; the same rule as `cuddle_merge_topk`).

Two variants:

* `ShardedSearcher` (simple): each rank selects its own top rerank_k candidates among its leaves,
  so up to world * rerank_k candidates are reranked in total: recall is >= the single-index recall,
  and the result equals the single-index search exactly whenever rerank_k covers every candidate.
  One collective per batch (the all-gather).

* `ExactShardedSearcher`: reproduces the single-index result bit for bit. The selection of the
  rerank_k best candidates is a property of all candidates of a query, so the ranks agree on it
  with two small all-reduces before reranking: the per-query Hamming histograms ([nq, dim+1] u32)
  give the global threshold and tie budget, the per-leaf-slot tie counts ([nq, leaf_stride] u32)
  decide which ties at the threshold are inside the budget. Every rank then reranks only its
  members of the global selection (rerank_k candidates in total, not world * rerank_k) and the
  all-gathered lists are merged by (distance, global phase-1 order).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from .flat_index import FlatIndex

PAD_ID = -1  # UINT64_MAX seen through int64


def merge_gathered(all_ids: torch.Tensor, all_dist: torch.Tensor, k: int):
    """all_ids/all_dist: [world, nq, k] (ascending per list, padded with id -1 / +inf).
    Returns (ids [nq,k], dist [nq,k], count [nq])."""
    world, nq, kk = all_dist.shape
    d = all_dist.permute(1, 0, 2).reshape(nq, world * kk)
    i = all_ids.permute(1, 0, 2).reshape(nq, world * kk)
    d = torch.where(i == PAD_ID, torch.full_like(d, float("inf")), d)
    sd, order = torch.sort(d, dim=1, stable=True)
    si = torch.gather(i, 1, order)
    sd, si = sd[:, :k].contiguous(), si[:, :k].contiguous()
    count = (si != PAD_ID).sum(dim=1).to(torch.int32)
    return si, sd, count


def _via_host(group) -> bool:
    """gloo moves CUDA tensors through the host on streams of its own; doing the staging here instead
    (plain copies on the current stream) keeps a rehearsal of the N > 1 flow on a shared GPU deterministic.
    RCCL ("nccl") works on the device buffers directly."""
    return dist.get_backend(group) == "gloo"


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    if t.is_cuda and _via_host(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather(out: torch.Tensor, mine: torch.Tensor, group) -> None:
    """out[r] = rank r's `mine` (out: [world, *mine.shape])."""
    mine = mine.contiguous()
    if mine.is_cuda and _via_host(group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather(list(h.unbind(0)), mine.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather(list(out.unbind(0)), mine, group=group)


class CommClock:
    """Time and bytes of every collective of the sharded protocol, by name: HIP events on the current stream around
    the call for RCCL (the call makes the stream wait for the collective), the host clock for the staged gloo
    rehearsals. `summary()` gives, per name, calls, milliseconds per call and bytes this rank sends per call."""

    def __init__(self):
        self.pending = []   # (name, start event, end event)
        self.ms = {}
        self.calls = {}
        self.bytes = {}

    def run(self, name: str, nbytes: int, fn, tensor: torch.Tensor, group):
        self.calls[name] = self.calls.get(name, 0) + 1
        self.bytes[name] = nbytes
        if tensor.is_cuda and not _via_host(group):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            self.pending.append((name, e0, e1))
        else:
            import time
            if tensor.is_cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            self.ms[name] = self.ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    def summary(self) -> dict:
        if self.pending:
            torch.cuda.synchronize()
            for name, e0, e1 in self.pending:
                self.ms[name] = self.ms.get(name, 0.0) + e0.elapsed_time(e1)
            self.pending = []
        return {n: {"calls": c, "ms_per_call": self.ms.get(n, 0.0) / c, "bytes_per_call": self.bytes[n]} for n, c in self.calls.items()}

    def reset(self):
        self.summary()
        self.ms, self.calls, self.bytes = {}, {}, {}


class ShardedSearcher:
    """search_batch over `world` leaf shards. `local_search(queries, k, probes, rerank_factor)` must
    return (ids int64 [nq,k], dist f32 [nq,k], count) tensors on the collective's device; on the
    GPU it is `DeviceIndex.search_batch_device` of this rank's shard."""

    def __init__(self, local_search: Callable, group: Optional[dist.ProcessGroup] = None):
        self.local_search = local_search
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @staticmethod
    def shard_of(flat: FlatIndex, rank: int, world: int) -> FlatIndex:
        return flat.shard(rank, world)

    def search_batch(self, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int):
        ids, dist_, count = self.local_search(queries, k, probes_per_level, rerank_factor)
        if self.world == 1:
            return ids, dist_, count
        nq = ids.shape[0]
        all_ids = torch.empty((self.world, nq, k), dtype=ids.dtype, device=ids.device)
        all_dist = torch.empty((self.world, nq, k), dtype=dist_.dtype, device=dist_.device)
        _all_gather(all_ids, ids, self.group)
        _all_gather(all_dist, dist_, self.group)
        return merge_gathered(all_ids, all_dist, k)


def merge_gathered_ordered(all_ids: torch.Tensor, all_dist: torch.Tensor, all_order: torch.Tensor, k: int):
    """all_*: [world, nq, k]. Ascending by (distance, order), the rule of cuddle_merge_topk_ordered.
    Returns (ids [nq,k], dist [nq,k], count [nq])."""
    world, nq, kk = all_dist.shape
    if all_dist.is_cuda and kk == k and world * k <= 1024:  # one kernel (cuddle_merge_topk_ordered_device)
        from . import _ffi
        ids = torch.empty((nq, k), dtype=torch.int64, device=all_dist.device)
        dd = torch.empty((nq, k), dtype=torch.float32, device=all_dist.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=all_dist.device)
        ai, ad, ao = all_ids.contiguous(), all_dist.contiguous(), all_order.contiguous()
        _ffi.check(_ffi.lib().cuddle_merge_topk_ordered_device(
            ai.data_ptr(), ad.data_ptr(), ao.data_ptr(), world, nq, k, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(),
            torch.cuda.current_stream(all_dist.device).cuda_stream))
        return ids, dd, cnt
    d = all_dist.permute(1, 0, 2).reshape(nq, world * kk)
    i = all_ids.permute(1, 0, 2).reshape(nq, world * kk)
    o = all_order.permute(1, 0, 2).reshape(nq, world * kk)
    pad = i == PAD_ID
    d = torch.where(pad, torch.full_like(d, float("inf")), d)
    # order keys are u64 seen through int64: flipping the top bit makes the signed order the unsigned one; pads sort last
    o = torch.where(pad, torch.full_like(o, torch.iinfo(torch.int64).max), o ^ torch.iinfo(torch.int64).min)
    _, p1 = torch.sort(o, dim=1, stable=True)          # secondary key first ...
    d1, i1 = torch.gather(d, 1, p1), torch.gather(i, 1, p1)
    sd, p2 = torch.sort(d1, dim=1, stable=True)        # ... then a stable sort by the primary key
    si = torch.gather(i1, 1, p2)
    sd, si = sd[:, :k].contiguous(), si[:, :k].contiguous()
    count = (si != PAD_ID).sum(dim=1).to(torch.int32)
    return si, sd, count


def merge_gathered_packed(everyone: torch.Tensor, nq: int, k: int):
    """everyone: [world, words] int64, each row one shard's packed result (DeviceIndex.packed_len:
    ids, order keys, f32 distances back to back). Same result as merge_gathered_ordered."""
    world, words = everyone.shape
    n = nq * k
    if everyone.is_cuda and world * k <= 1024:  # the merge kernel reads the messages in place
        from . import _ffi
        ids = torch.empty((nq, k), dtype=torch.int64, device=everyone.device)
        dd = torch.empty((nq, k), dtype=torch.float32, device=everyone.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=everyone.device)
        base = everyone.data_ptr()
        _ffi.check(_ffi.lib().cuddle_merge_topk_ordered_device(
            base, base + 2 * n * 8, base + n * 8, world, nq, k, words * 8, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(),
            torch.cuda.current_stream(everyone.device).cuda_stream))
        return ids, dd, cnt
    all_ids = everyone[:, :n].reshape(world, nq, k)
    all_order = everyone[:, n:2 * n].reshape(world, nq, k)
    all_dist = everyone[:, 2 * n:].contiguous().view(torch.float32)[:, :n].reshape(world, nq, k)
    return merge_gathered_ordered(all_ids, all_dist, all_order, k)


class ExactShardedSearcher:
    """Bit-exact leaf-sharded search. `engine` is this rank's shard and provides
    shard_scan(queries, k, probes, rf) -> hist, shard_ties(global_hist) -> ties,
    shard_finish(queries, global_ties) -> (ids, dist, order, count) as tensors on the collective's
    device (`DeviceIndex` on the GPU)."""

    COARSE_BIN = 16        # Hamming values per coarse bin of the two-step histogram exchange
    COARSE_MIN_BINS = 256  # "auto": histograms of at least this many bins travel coarse first

    def __init__(self, engine, group: Optional[dist.ProcessGroup] = None, shard_routing: bool = True,
                 hist_exchange: str = "auto", clock: Optional[CommClock] = None):
        """hist_exchange: how the ranks agree on the Hamming threshold. "full": one all-reduce of the [nq, dim + 1]
        histograms (the protocol's one message that grows with the dimension: 31 MB for 8000 queries of 960-d).
        "coarse_fine": an all-reduce of the histograms in bins of 16 values, then one of the 16 values of the bin
        the threshold falls in - the same threshold and tie budget from two small messages. "auto": coarse_fine
        from 256 bins on. clock: a CommClock that times every collective (None: no timing, no events)."""
        assert hist_exchange in ("auto", "full", "coarse_fine")
        self.hist_exchange = hist_exchange
        self.clock = clock
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the tree is replicated, so routing gives the same leaf lists on every rank: each rank routes
        # 1/world of the batch and the lists are all-gathered, instead of every rank routing all of it
        self.shard_routing = shard_routing and self.world > 1 and hasattr(engine, "route_device")
        # engines that can write their outputs into caller-given message buffers (DeviceIndex) save the packing copies
        self.packed = hasattr(engine, "packed_len")

    def _gather(self, name: str, out: torch.Tensor, mine: torch.Tensor):
        if self.clock is None:
            _all_gather(out, mine, self.group)
        else:
            self.clock.run(name, mine.numel() * mine.element_size(), lambda: _all_gather(out, mine, self.group), mine, self.group)

    def _reduce(self, name: str, t: torch.Tensor):
        if self.clock is None:
            _all_reduce_sum(t, self.group)
        else:
            self.clock.run(name, t.numel() * t.element_size(), lambda: _all_reduce_sum(t, self.group), t, self.group)

    def _global_histogram(self, hist: torch.Tensor, rerank_k: int) -> torch.Tensor:
        """The sum of the ranks' [nq, bins] histograms, or a stand-in that gives cuddle_shard_ties the same threshold
        and tie budget: those depend on the running sums only, so everything below the coarse bin in which the
        rerank_k-th candidate falls may sit in one earlier bin, and what lies above it is never looked at."""
        nq, bins = hist.shape
        mode = self.hist_exchange
        if mode == "auto":
            mode = "coarse_fine" if bins >= self.COARSE_MIN_BINS else "full"
        if mode == "full" or self.world == 1:
            if self.world > 1:
                self._reduce("hist_allreduce", hist)
            return hist
        B = self.COARSE_BIN
        nb = (bins + B - 1) // B
        padded = torch.zeros((nq, nb * B), dtype=hist.dtype, device=hist.device)
        padded[:, :bins] = hist
        local = padded.view(nq, nb, B)
        coarse = local.sum(dim=2, dtype=torch.int64).to(torch.int32).contiguous()
        self._reduce("hist_coarse_allreduce", coarse)
        cum = coarse.to(torch.int64).cumsum(dim=1)
        reached = cum >= rerank_k
        # the first coarse bin whose running sum reaches rerank_k; none (fewer candidates than that): the last bin
        c0 = torch.where(reached.any(dim=1), reached.to(torch.int8).argmax(dim=1), torch.full((nq,), nb - 1, device=hist.device))
        rows = torch.arange(nq, device=hist.device)
        below = (cum[rows, c0] - coarse[rows, c0].to(torch.int64)).to(torch.int32)
        fine = local[rows, c0].contiguous()  # [nq, B]: this rank's counts of the bin's values
        self._reduce("hist_fine_allreduce", fine)
        out = torch.zeros((nq, nb * B), dtype=hist.dtype, device=hist.device)
        cols = c0[:, None] * B + torch.arange(B, device=hist.device)[None, :]
        out.scatter_(1, cols, fine)
        has_below = c0 > 0
        out[rows[has_below], (c0[has_below] * B - 1)] = below[has_below]
        return out[:, :bins].contiguous()

    def _route_sharded(self, queries: torch.Tensor, probes_per_level: int):
        nq = queries.shape[0]
        per = (nq + self.world - 1) // self.world
        q0 = min(nq, self.rank * per)
        q1 = min(nq, q0 + per)
        if self.packed:  # the engine writes the lists straight into the message: [per][stride + 1], last column = counts
            stride = self.engine.leaf_stride(probes_per_level)
            mine = torch.empty((per, stride + 1), dtype=torch.int32, device=queries.device)
            _, count = self.engine.route_device(queries[q0:q1], probes_per_level, out=mine)
            mine[:q1 - q0, stride] = count
        else:
            leaves, count = self.engine.route_device(queries[q0:q1].contiguous(), probes_per_level)
            stride = leaves.shape[1]
            mine = torch.zeros((per, stride + 1), dtype=leaves.dtype, device=leaves.device)
            mine[:q1 - q0, :stride] = leaves
            mine[:q1 - q0, stride] = count
        everyone = torch.empty((self.world, per, stride + 1), dtype=mine.dtype, device=mine.device)
        self._gather("leaf_gather", everyone, mine)
        everyone = everyone.reshape(self.world * per, stride + 1)[:nq]  # rows past a rank's slice lie past nq
        if self.packed:
            return everyone[:, :stride], everyone[:, stride].contiguous()  # the lists stay where the gather put them
        return everyone[:, :stride].contiguous(), everyone[:, stride].contiguous()

    MAX_ROUND = 4096  # queries per round of the protocol (the library's limit for one sharded batch)

    @staticmethod
    def _rerank_k(k: int, rerank_factor: int) -> int:
        return min(k * rerank_factor, 10000)  # search() caps the selection (hierarchical.rs:447)

    def search_batch(self, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int):
        """Larger batches run as equal rounds of at most MAX_ROUND queries, one after the other (every rank cuts
        the batch the same way: the cut depends on the batch size alone)."""
        nq = queries.shape[0]
        if nq <= self.MAX_ROUND:
            return self._search_round(queries, k, probes_per_level, rerank_factor)
        rounds = (nq + self.MAX_ROUND - 1) // self.MAX_ROUND
        per = (nq + rounds - 1) // rounds
        parts = [self._search_round(queries[q0:q0 + per], k, probes_per_level, rerank_factor) for q0 in range(0, nq, per)]
        return tuple(torch.cat([p[i] for p in parts]) for i in range(3))

    def _search_round(self, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int):
        leaves = count = None
        if self.shard_routing:
            leaves, count = self._route_sharded(queries, probes_per_level)
        if getattr(self.engine, "selects_by_keys", False):
            # RaBitQ codes: every rank's own selection, named by keys that order candidates as the unsharded
            # selection does; one all-gather of the keys, then each rank keeps what lies under the global threshold
            keys = self.engine.shard_scan_keys(queries, k, probes_per_level, rerank_factor, leaves=leaves, leaf_count=count)
            all_keys = torch.empty((self.world,) + tuple(keys.shape), dtype=keys.dtype, device=keys.device)
            if self.world > 1:
                self._gather("key_gather", all_keys, keys)
            else:
                all_keys[0] = keys
            finish = lambda **kw: self.engine.shard_finish_keys(queries, all_keys, **kw)  # noqa: E731
        else:
            hist = self.engine.shard_scan(queries, k, probes_per_level, rerank_factor, leaves=leaves, leaf_count=count) \
                if leaves is not None else self.engine.shard_scan(queries, k, probes_per_level, rerank_factor)
            hist = self._global_histogram(hist, self._rerank_k(k, rerank_factor))
            ties = self.engine.shard_ties(hist)
            if self.world > 1:
                self._reduce("tie_allreduce", ties)
            finish = lambda **kw: self.engine.shard_finish(queries, ties, **kw)  # noqa: E731
        if self.packed and self.world > 1:
            nq = queries.shape[0]
            words = self.engine.packed_len(nq, k)
            mine = torch.empty((words,), dtype=torch.int64, device=queries.device)
            finish(packed=mine)
            everyone = torch.empty((self.world, words), dtype=torch.int64, device=queries.device)
            self._gather("result_gather", everyone, mine)
            return merge_gathered_packed(everyone, nq, k)
        ids, dist_, order, count = finish()
        if self.world == 1:
            return ids, dist_, count
        nq = ids.shape[0]
        # one message per rank: [3][nq][k] int64 = ids, order keys, distance bits
        mine = torch.stack((ids, order, dist_.contiguous().view(torch.int32).to(torch.int64)))
        everyone = torch.empty((self.world, 3, nq, k), dtype=torch.int64, device=ids.device)
        self._gather("result_gather", everyone, mine)
        all_dist = everyone[:, 2].to(torch.int32).view(torch.float32)
        return merge_gathered_ordered(everyone[:, 0], all_dist, everyone[:, 1], k)


def exact_search_in_process(engines, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int,
                            shard_routing: bool = False, packed: bool = False):
    """The ExactShardedSearcher protocol over several shards held by ONE process (sums instead of
    all-reduces, concatenation instead of the all-gathers): for tests and for single-GPU rehearsals
    of an N-way split. packed: the engines write into message buffers as they do between ranks."""
    nq, world = queries.shape[0], len(engines)
    if shard_routing:
        per = (nq + world - 1) // world
        if packed:
            stride = engines[0].leaf_stride(probes_per_level)
            everyone = torch.empty((world, per, stride + 1), dtype=torch.int32, device=queries.device)
            for r, e in enumerate(engines):
                q0, q1 = min(nq, r * per), min(nq, (r + 1) * per)
                _, c = e.route_device(queries[q0:q1], probes_per_level, out=everyone[r])
                everyone[r, :q1 - q0, stride] = c
            everyone = everyone.reshape(world * per, stride + 1)[:nq]
            leaves, count = everyone[:, :stride], everyone[:, stride].contiguous()
        else:
            parts = [e.route_device(queries[min(nq, r * per):min(nq, (r + 1) * per)].contiguous(), probes_per_level)
                     for r, e in enumerate(engines)]
            leaves = torch.cat([p[0] for p in parts]).contiguous()
            count = torch.cat([p[1] for p in parts]).contiguous()
        lkw = dict(leaves=leaves, leaf_count=count)
    else:
        lkw = {}
    if getattr(engines[0], "selects_by_keys", False):  # RaBitQ codes: the shards' selection keys, concatenated
        all_keys = torch.stack([e.shard_scan_keys(queries, k, probes_per_level, rerank_factor, **lkw) for e in engines]).contiguous()
        finish = lambda e, **kw: e.shard_finish_keys(queries, all_keys, **kw)  # noqa: E731
    else:
        hists = [e.shard_scan(queries, k, probes_per_level, rerank_factor, **lkw) for e in engines]
        ghist = torch.stack(hists).sum(dim=0).to(torch.int32).contiguous()
        ties = [e.shard_ties(ghist) for e in engines]
        gties = torch.stack(ties).sum(dim=0).to(torch.int32).contiguous()
        finish = lambda e, **kw: e.shard_finish(queries, gties, **kw)  # noqa: E731
    if packed:
        words = engines[0].packed_len(nq, k)
        everyone = torch.empty((world, words), dtype=torch.int64, device=queries.device)
        for r, e in enumerate(engines):
            finish(e, packed=everyone[r])
        return merge_gathered_packed(everyone, nq, k)
    outs = [finish(e) for e in engines]
    all_ids = torch.stack([o[0] for o in outs])
    all_dist = torch.stack([o[1] for o in outs])
    all_order = torch.stack([o[2] for o in outs])
    return merge_gathered_ordered(all_ids, all_dist, all_order, k)


class RcclShardedSearcher:
    """The exact sharded search with the collectives inside the library (cuddle_comm_* /
    cuddle_sharded_search_batch: RCCL called from C, no torch.distributed on the data path).
    `unique_id` comes from `RcclShardedSearcher.make_unique_id()` on rank 0 and reaches the other
    ranks through any channel the host has (here typically torch.distributed's store)."""

    def __init__(self, dev_index, rank: int, world: int, unique_id: bytes):
        import ctypes as C

        from . import _ffi
        self._ffi, self._C = _ffi, C
        self.dev = dev_index
        self.rank, self.world = rank, world
        self._comm = C.c_void_p(None)
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _ffi.check(_ffi.lib().cuddle_comm_init(rank, world, C.cast(buf, C.c_void_p), dev_index.device_id, C.byref(self._comm)))

    @staticmethod
    def make_unique_id() -> bytes:
        import ctypes as C

        from . import _ffi
        buf = (C.c_uint8 * 128)()
        _ffi.check(_ffi.lib().cuddle_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def close(self) -> None:
        if self._comm is not None and self._comm.value:
            self._ffi.lib().cuddle_comm_destroy(self._comm)
            self._comm = self._C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_batch(self, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                     out_ids=None, out_dist=None, out_count=None):
        C = self._C
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq, dim = queries.shape
        dev = queries.device
        ids = out_ids if out_ids is not None else torch.empty((nq, k), dtype=torch.int64, device=dev)
        dd = out_dist if out_dist is not None else torch.empty((nq, k), dtype=torch.float32, device=dev)
        cnt = out_count if out_count is not None else torch.empty((nq,), dtype=torch.int32, device=dev)
        self._ffi.check(self._ffi.lib().cuddle_sharded_search_batch(
            self.dev._handle, self._comm, C.c_void_p(queries.data_ptr()), nq, dim, k, probes_per_level, rerank_factor, flags,
            C.c_void_p(ids.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(cnt.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return ids, dd, cnt
