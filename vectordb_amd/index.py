"""Host-side mirror of the reference's search interface, on top of the C ABI.

Same names, argument meaning and error behaviour as
`ClusteredIndex::{build, search, search_with_stats}` (/root/reference/src/index/hierarchical.rs:87,330,831)
and `ClusteredIndexWithRaBitQ` (src/index/hierarchical_with_rabitq.rs:84,329,809); `search_batch`
is the batched form the reference only sketches (benches/profile_bench.rs:286-296). All compute
runs in libcuddle_hip.so on the GPU; there is no CPU path behind these classes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .flat_index import FlatIndex

UINT64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


@dataclass
class SearchStats:
    """SearchStats (hierarchical.rs:36-45), one per query, field for field."""
    total_leaves: int
    leaves_searched: int
    leaves_searched_ids: List[int]   # accumulation order (hierarchical.rs:902, :917)
    total_vectors: int
    vectors_scanned_binary: int
    vectors_reranked_full: int
    tree_depth: int                  # depth of the deepest node (the reference's max_depth, :851)
    probes_per_level: List[int]      # frontier size at every level, ending with the empty frontier (:861)


class DeviceIndex:
    """Owns one `cuddle_index` handle (device-resident index + workspace)."""

    def __init__(self, flat: Optional[FlatIndex], device_id: int = 0, desc: Optional[_ffi.IndexDesc] = None,
                 keepalive=None):
        """From a host FlatIndex, or from a ready `desc` (e.g. with CUDDLE_DESC_DEVICE_ARRAYS: see
        vectordb_amd/synthetic.py) whose arrays `keepalive` keeps referenced."""
        self._lib = _ffi.lib()
        self._handle = C.c_void_p(None)
        self._keepalive = keepalive
        if desc is None:
            self.flat = flat.normalised()
            desc = self.flat.desc()
        else:
            self.flat = flat
        _ffi.check(self._lib.cuddle_index_create(C.byref(desc), int(device_id), C.byref(self._handle)))
        self.device_id = int(device_id)
        self.dim = int(desc.dim)
        self.code_kind = int(desc.code_kind)
        self.levels = int(self._lib.cuddle_index_levels(self._handle))
        self.max_leaf_size = int(self._lib.cuddle_index_max_leaf_size(self._handle))
        self.max_children = int(self._lib.cuddle_index_max_children(self._handle))

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.cuddle_index_destroy(self._handle)
            self._handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(self._lib.cuddle_index_device_bytes(self._handle))

    def reserve(self, max_queries: int, max_k: int, max_probes: int, max_rerank_k: int) -> None:
        _ffi.check(self._lib.cuddle_index_reserve(self._handle, max_queries, max_k, max_probes, max_rerank_k))

    # ---- the hot path -------------------------------------------------------------------
    def search_batch(self, queries: np.ndarray, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                     want_stats: bool = False):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError("queries must be [n_queries, dim]")
        nq, dim = q.shape
        ids = np.empty((nq, max(k, 0)), dtype=np.uint64)
        dist = np.empty((nq, max(k, 0)), dtype=np.float32)
        count = np.empty(nq, dtype=np.uint32)
        st = _ffi.Stats()
        _ffi.check(self._lib.cuddle_search_batch(self._handle, _ffi.ptr(q), nq, dim, k, probes_per_level, rerank_factor,
                                                 flags, _ffi.ptr(ids), _ffi.ptr(dist), _ffi.ptr(count),
                                                 C.byref(st) if want_stats else None))
        return (ids, dist, count, st.as_dict()) if want_stats else (ids, dist, count)

    def search_batch_device(self, d_queries, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                            out_ids=None, out_dist=None, out_count=None, stream: Optional[int] = None):
        """`d_queries`: float32 CUDA tensor [nq, dim] on this index's device. Enqueues on `stream`
        (a hipStream_t handle; default: torch's current stream). Returns (ids int64 view of the u64
        ids, dist, count) tensors; valid once the stream is synchronised."""
        import torch

        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq, dim = d_queries.shape
        dev = d_queries.device
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        if out_dist is None:
            out_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        if out_count is None:
            out_count = torch.empty((nq,), dtype=torch.int32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _ffi.check(self._lib.cuddle_search_batch_device(
            self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim, k, probes_per_level, rerank_factor, flags,
            C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_dist.data_ptr()), C.c_void_p(out_count.data_ptr()),
            C.c_void_p(stream)))
        return out_ids, out_dist, out_count

    def search_batch_device_noalloc(self, d_queries, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                                    out_ids=None, out_dist=None, out_count=None, stream: Optional[int] = None):
        """search_batch_device through cuddle_search_batch_device_noalloc: the library lays its work
        space out in its fixed per-GPU array and allocates nothing (the outputs are the caller's)."""
        import torch

        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq, dim = d_queries.shape
        assert out_ids is not None and out_dist is not None and out_count is not None, "the caller owns the outputs"
        if stream is None:
            stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        _ffi.check(self._lib.cuddle_search_batch_device_noalloc(
            self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim, k, probes_per_level, rerank_factor, flags,
            C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_dist.data_ptr()), C.c_void_p(out_count.data_ptr()),
            C.c_void_p(stream)))
        return out_ids, out_dist, out_count

    def search_batch_with_stats(self, queries: np.ndarray, k: int, probes: int, rerank_factor: int):
        """search_with_stats (hierarchical.rs:831-938) for every row of `queries`: (ids, dist, count,
        leaves [nq, leaf_stride], leaf_count, level_nodes [nq, levels + 1], level_count, scanned, reranked)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = q.shape
        ids = np.empty((nq, k), dtype=np.uint64)
        dist = np.empty((nq, k), dtype=np.float32)
        count = np.zeros(nq, dtype=np.uint32)
        lstride = self.leaf_stride(probes)
        leaves = np.zeros((nq, lstride), dtype=np.uint32)
        lcount = np.zeros(nq, dtype=np.uint32)
        level_nodes = np.zeros((nq, self.levels + 1), dtype=np.uint32)
        level_count = np.zeros(nq, dtype=np.uint32)
        scanned = np.zeros(nq, dtype=np.uint32)
        reranked = np.zeros(nq, dtype=np.uint32)
        _ffi.check(self._lib.cuddle_search_batch_with_stats(
            self._handle, _ffi.ptr(q), nq, dim, k, probes, rerank_factor, _ffi.ptr(ids), _ffi.ptr(dist), _ffi.ptr(count),
            _ffi.ptr(leaves), lstride, _ffi.ptr(lcount), _ffi.ptr(level_nodes), self.levels + 1, _ffi.ptr(level_count),
            _ffi.ptr(scanned), _ffi.ptr(reranked)))
        return ids, dist, count, leaves, lcount, level_nodes, level_count, scanned, reranked

    # ---- leaf-sharded exact search: the three steps between which the shards exchange counts ----
    def shard_scan(self, d_queries, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                   stream: Optional[int] = None, leaves=None, leaf_count=None):
        """Step 1 on this shard: returns the shard's Hamming histogram, int32 CUDA tensor
        [nq, dim + 1], to be summed over the shards."""
        import torch

        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq, dim = d_queries.shape
        if stream is None:
            stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        hist = torch.empty((nq, dim + 1), dtype=torch.int32, device=d_queries.device)
        stride = C.c_uint32(0)
        if leaves is None:
            _ffi.check(self._lib.cuddle_shard_scan(self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim, k, probes_per_level,
                                                   rerank_factor, flags, C.c_void_p(hist.data_ptr()), C.byref(stride),
                                                   C.c_void_p(stream)))
        else:  # routing done elsewhere (route_device on slices of the batch, all-gathered)
            assert leaves.is_cuda and leaves.dtype == torch.int32 and leaves.stride(1) == 1 and leaves.shape[0] == nq
            assert leaf_count.is_cuda and leaf_count.dtype == torch.int32 and leaf_count.is_contiguous()
            _ffi.check(self._lib.cuddle_shard_scan_leaves(self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim, k,
                                                          probes_per_level, rerank_factor, flags,
                                                          C.c_void_p(leaves.data_ptr()), leaves.stride(0),
                                                          C.c_void_p(leaf_count.data_ptr()), C.c_void_p(hist.data_ptr()),
                                                          C.byref(stride), C.c_void_p(stream)))
        self._shard = (nq, k, int(stride.value))
        return hist

    def leaf_stride(self, probes_per_level: int) -> int:
        """Capacity of a query's leaf list for this probes_per_level (a property of the tree)."""
        out = C.c_uint32(0)
        _ffi.check(self._lib.cuddle_index_leaf_stride(self._handle, probes_per_level, C.byref(out)))
        return int(out.value)

    def route_device(self, d_queries, probes_per_level: int, flags: int = 0, stream: Optional[int] = None, out=None):
        """Routing only, on the device: (leaves int32 [nq, leaf_stride], leaf_count int32 [nq]).
        out: an int32 CUDA tensor [>= nq, >= leaf_stride] (unit stride along a row) that receives the
        lists in its first leaf_stride columns; `leaves` is then a view of it."""
        import torch

        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq, dim = d_queries.shape
        if stream is None:
            stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        stride = self.leaf_stride(probes_per_level)
        if out is None:
            leaves = torch.zeros((nq, stride), dtype=torch.int32, device=d_queries.device)
        else:
            assert out.is_cuda and out.dtype == torch.int32 and out.stride(1) == 1 and out.shape[0] >= nq and out.shape[1] >= stride
            leaves = out[:nq, :stride]
        count = torch.zeros((nq,), dtype=torch.int32, device=d_queries.device)
        if nq:
            _ffi.check(self._lib.cuddle_route_batch_device(self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim,
                                                           probes_per_level, flags, C.c_void_p(leaves.data_ptr()),
                                                           leaves.stride(0), C.c_void_p(count.data_ptr()), C.c_void_p(stream)))
        return leaves, count

    def shard_ties(self, global_hist, stream: Optional[int] = None):
        """Step 2: `global_hist` = the histograms summed over all shards. Returns this shard's tie
        counts per leaf slot, int32 CUDA tensor [nq, leaf_stride], to be summed likewise."""
        import torch

        nq, _, stride = self._shard
        if stream is None:
            stream = torch.cuda.current_stream(global_hist.device).cuda_stream
        assert global_hist.is_contiguous() and global_hist.dtype == torch.int32
        ties = torch.empty((nq, stride), dtype=torch.int32, device=global_hist.device)
        _ffi.check(self._lib.cuddle_shard_ties(self._handle, C.c_void_p(global_hist.data_ptr()), C.c_void_p(ties.data_ptr()),
                                               C.c_void_p(stream)))
        return ties

    @staticmethod
    def packed_len(nq: int, k: int) -> int:
        """int64 words of one shard's packed result: ids [nq*k], order keys [nq*k], distances (f32, two per word)."""
        return 2 * nq * k + (nq * k + 1) // 2

    def shard_finish(self, d_queries, global_ties, stream: Optional[int] = None, packed=None):
        """Step 3: returns (ids int64 view of u64, dist f32, order int64 view of u64, count int32),
        each [nq, k] / [nq]: this shard's top-k of the global selection. packed: an int64 CUDA tensor
        of packed_len(nq, k) words that receives ids, order keys and distances back to back (one
        message for the all-gather); the returned tensors are then views of it."""
        import torch

        nq, k, _ = self._shard
        dev = d_queries.device
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        assert global_ties.is_contiguous() and global_ties.dtype == torch.int32
        if packed is None:
            ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
            order = torch.empty((nq, k), dtype=torch.int64, device=dev)
        else:
            assert packed.is_cuda and packed.dtype == torch.int64 and packed.is_contiguous() and packed.numel() >= self.packed_len(nq, k)
            n = nq * k
            ids, order = packed[:n].view(nq, k), packed[n:2 * n].view(nq, k)
            dist = packed[2 * n:].view(torch.float32)[:n].view(nq, k)
        count = torch.empty((nq,), dtype=torch.int32, device=dev)
        _ffi.check(self._lib.cuddle_shard_finish(self._handle, C.c_void_p(d_queries.data_ptr()),
                                                 C.c_void_p(global_ties.data_ptr()), C.c_void_p(ids.data_ptr()),
                                                 C.c_void_p(dist.data_ptr()), C.c_void_p(order.data_ptr()),
                                                 C.c_void_p(count.data_ptr()), C.c_void_p(stream)))
        return ids, dist, order, count

    # ---- the same for RaBitQ codes: the shards exchange the keys of their own selections ----
    @property
    def selects_by_keys(self) -> bool:
        """RaBitQ codes: the exact sharded protocol exchanges selection keys instead of histograms."""
        return self.code_kind == _ffi.CODE_RABITQ

    def shard_scan_keys(self, d_queries, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0,
                        stream: Optional[int] = None, leaves=None, leaf_count=None):
        """Step 1 for RaBitQ codes (cuddle_shard_scan_rabitq): this shard's selection keys, int64 view of u64
        [nq, rerank_k], ascending, padded with -1 (UINT64_MAX); to be all-gathered."""
        import torch

        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq, dim = d_queries.shape
        if stream is None:
            stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        r = C.c_uint32(0)
        _ffi.check(self._lib.cuddle_rerank_k(k, rerank_factor, flags, C.byref(r)))
        keys = torch.empty((nq, int(r.value)), dtype=torch.int64, device=d_queries.device)
        lp, ls, lc = None, 0, None
        if leaves is not None:
            assert leaves.is_cuda and leaves.dtype == torch.int32 and leaves.stride(1) == 1 and leaves.shape[0] == nq
            assert leaf_count.is_cuda and leaf_count.dtype == torch.int32 and leaf_count.is_contiguous()
            lp, ls, lc = C.c_void_p(leaves.data_ptr()), leaves.stride(0), C.c_void_p(leaf_count.data_ptr())
        _ffi.check(self._lib.cuddle_shard_scan_rabitq(self._handle, C.c_void_p(d_queries.data_ptr()), nq, dim, k, probes_per_level,
                                                      rerank_factor, flags, lp, ls, lc, C.c_void_p(keys.data_ptr()), C.c_void_p(stream)))
        self._shard = (nq, k, 0)
        return keys

    def shard_finish_keys(self, d_queries, all_keys, stream: Optional[int] = None, packed=None):
        """Step 2 for RaBitQ codes (cuddle_shard_finish_rabitq): all_keys int64 [n_shards, nq, rerank_k] = every
        shard's keys. Returns what shard_finish returns."""
        import torch

        nq, k, _ = self._shard
        dev = d_queries.device
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        assert all_keys.is_cuda and all_keys.dtype == torch.int64 and all_keys.is_contiguous() and all_keys.shape[1] == nq
        if packed is None:
            ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
            order = torch.empty((nq, k), dtype=torch.int64, device=dev)
        else:
            assert packed.is_cuda and packed.dtype == torch.int64 and packed.is_contiguous() and packed.numel() >= self.packed_len(nq, k)
            n = nq * k
            ids, order = packed[:n].view(nq, k), packed[n:2 * n].view(nq, k)
            dist = packed[2 * n:].view(torch.float32)[:n].view(nq, k)
        count = torch.empty((nq,), dtype=torch.int32, device=dev)
        _ffi.check(self._lib.cuddle_shard_finish_rabitq(self._handle, C.c_void_p(d_queries.data_ptr()), C.c_void_p(all_keys.data_ptr()),
                                                        all_keys.shape[0], C.c_void_p(ids.data_ptr()), C.c_void_p(dist.data_ptr()),
                                                        C.c_void_p(order.data_ptr()), C.c_void_p(count.data_ptr()), C.c_void_p(stream)))
        return ids, dist, order, count

    def last_stats(self) -> dict:
        st = _ffi.Stats()
        _ffi.check(self._lib.cuddle_last_stats(self._handle, C.byref(st)))
        return st.as_dict()

    def stage_times(self) -> dict:
        """Device milliseconds per stage summed over the timed searches since the last call, plus
        "searches" = how many searches that covers."""
        ms = np.zeros(len(_ffi.STAGE_NAMES), dtype=np.float32)
        n = np.zeros(1, dtype=np.uint32)
        _ffi.check(self._lib.cuddle_stage_times(self._handle, _ffi.ptr(ms), _ffi.ptr(n)))
        out = dict(zip(_ffi.STAGE_NAMES, (float(x) for x in ms)))
        out["searches"] = int(n[0])
        return out

    # ---- stage entry points (parity tests) ------------------------------------------------
    def route_batch(self, queries: np.ndarray, probes_per_level: int, flags: int = 0):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = q.shape
        stride = max(1, self.leaf_stride(probes_per_level))
        leaves = np.zeros((nq, stride), dtype=np.uint32)
        count = np.zeros(nq, dtype=np.uint32)
        _ffi.check(self._lib.cuddle_route_batch(self._handle, _ffi.ptr(q), nq, dim, probes_per_level, flags,
                                                _ffi.ptr(leaves), stride, _ffi.ptr(count)))
        return leaves, count

    def route_filter_error(self, queries: np.ndarray, probes_per_level: int, level: int):
        """(max err / bound, max err) of the routing filter's approximate distances on one tree level."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = q.shape
        ratio = np.zeros(1, dtype=np.float32)
        err = np.zeros(1, dtype=np.float32)
        _ffi.check(self._lib.cuddle_route_filter_error(self._handle, _ffi.ptr(q), nq, dim, probes_per_level, level,
                                                       _ffi.ptr(ratio), _ffi.ptr(err)))
        return float(ratio[0]), float(err[0])

    def scan_mode(self, n_queries: int, probes_per_level: int) -> str:
        """How a batch of this size scans its leaves: "per_query" or "leaf_by_leaf" (cuddle_scan_mode)."""
        out = C.c_uint32(0)
        _ffi.check(self._lib.cuddle_scan_mode(self._handle, n_queries, probes_per_level, C.byref(out)))
        return "leaf_by_leaf" if out.value else "per_query"

    def scan_read_ceiling(self, d_batches, k: int, probes_per_level: int, rerank_factor: int, reps: int = 5):
        """d_batches: float32 CUDA tensor [n_batches, nq, dim]. (read_ms, scan_ms, code_bytes): mean launch time of the
        kernel that only issues the Hamming scan's reads, of the scan kernel itself (both going round the batches
        `reps` times), and the code bytes one launch covers (cuddle_scan_read_ceiling)."""
        assert d_batches.is_cuda and d_batches.is_contiguous() and d_batches.dim() == 3
        nb, nq, dim = d_batches.shape
        a, b = np.zeros(1, dtype=np.float32), np.zeros(1, dtype=np.float32)
        n = np.zeros(1, dtype=np.uint64)
        _ffi.check(self._lib.cuddle_scan_read_ceiling(self._handle, C.c_void_p(d_batches.data_ptr()), nb, nq, dim, k,
                                                      probes_per_level, rerank_factor, reps, _ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(n)))
        return float(a[0]), float(b[0]), int(n[0])

    def scan_batch(self, queries: np.ndarray, leaves: np.ndarray, leaf_count: np.ndarray):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = q.shape
        leaves = np.ascontiguousarray(leaves, dtype=np.uint32)
        leaf_count = np.ascontiguousarray(leaf_count, dtype=np.uint32)
        sizes = (self.flat.vec_offsets[1:] - self.flat.vec_offsets[:-1]).astype(np.int64)
        cand = max(1, int(max((sizes[leaves[i, :leaf_count[i]]].sum() for i in range(nq)), default=1)))
        pos = np.zeros((nq, cand), dtype=np.uint64)
        key = np.zeros((nq, cand), dtype=np.uint32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _ffi.check(self._lib.cuddle_scan_batch(self._handle, _ffi.ptr(q), nq, dim, _ffi.ptr(leaves), leaves.shape[1],
                                               _ffi.ptr(leaf_count), _ffi.ptr(pos), _ffi.ptr(key), cand, _ffi.ptr(cnt)))
        return pos, key, cnt

    def search_leaves_batch(self, queries: np.ndarray, leaves: np.ndarray, leaf_count: np.ndarray, k: int,
                            rerank_factor: int, flags: int = 0):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = q.shape
        leaves = np.ascontiguousarray(leaves, dtype=np.uint32)
        leaf_count = np.ascontiguousarray(leaf_count, dtype=np.uint32)
        rk = k * rerank_factor if (flags & _ffi.SEARCH_STATS_VARIANT) else min(k * rerank_factor, 10000)
        sstride = max(1, rk)
        ids = np.empty((nq, k), dtype=np.uint64)
        dist = np.empty((nq, k), dtype=np.float32)
        count = np.empty(nq, dtype=np.uint32)
        spos = np.zeros((nq, sstride), dtype=np.uint64)
        skey = np.zeros((nq, sstride), dtype=np.uint32)
        sdist = np.zeros((nq, sstride), dtype=np.float32)
        scount = np.zeros(nq, dtype=np.uint32)
        _ffi.check(self._lib.cuddle_search_leaves_batch(
            self._handle, _ffi.ptr(q), nq, dim, _ffi.ptr(leaves), leaves.shape[1], _ffi.ptr(leaf_count), k, rerank_factor,
            flags, _ffi.ptr(ids), _ffi.ptr(dist), _ffi.ptr(count), _ffi.ptr(spos), _ffi.ptr(skey), _ffi.ptr(sdist),
            sstride, _ffi.ptr(scount)))
        return ids, dist, count, spos, skey, sdist, scount


class ClusteredIndex:
    """ClusteredIndex (hierarchical.rs:51-70): hierarchical k-means tree + binary codes."""

    code_kind = "binary"

    def __init__(self, flat: FlatIndex, device_id: int = 0):
        self.flat = flat
        self.dev = DeviceIndex(flat, device_id)
        self.dimension = flat.dim

    @classmethod
    def build(cls, vectors, vector_file: Optional[str] = None, branching_factor: int = 100, target_leaf_size: int = 100,
              metric="l2", max_iterations: int = 20, seed: int = 0, device_id: int = 0, build_device: Optional[str] = None):
        """ClusteredIndex::build (hierarchical.rs:87-174). `vector_file`, when given, receives the
        raw little-endian f32 rows exactly as MmapVectorStore::create writes them (mmap.rs:59-98)."""
        from .builder import build_index

        flat = build_index(vectors, branching_factor, target_leaf_size, metric, max_iterations, seed,
                           code_kind=cls.code_kind, device=build_device)
        if vector_file is not None:
            order = np.argsort(flat.ids, kind="stable")
            flat.vectors[order].astype("<f4").tofile(vector_file)
        return cls(flat, device_id)

    def __len__(self) -> int:
        return self.flat.n_vectors

    def search(self, query: Sequence[float], k: int, probes_per_level: int, rerank_factor: int) -> List[Tuple[int, float]]:
        """search (hierarchical.rs:330-336): [(vector index, distance)] ascending, at most k."""
        q = np.asarray(query, dtype=np.float32).reshape(1, -1)
        ids, dist, count = self.dev.search_batch(q, k, probes_per_level, rerank_factor)
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(int(count[0]))]

    def search_batch(self, queries: np.ndarray, k: int, probes_per_level: int, rerank_factor: int, flags: int = 0):
        """One search() per row; returns (ids [nq,k] u64 padded with UINT64_MAX, dist [nq,k] f32 padded
        with +inf, count [nq])."""
        return self.dev.search_batch(queries, k, probes_per_level, rerank_factor, flags)

    def search_with_stats(self, query: Sequence[float], k: int, probes: int, rerank_factor: int):
        """search_with_stats (hierarchical.rs:831-938)."""
        q = np.asarray(query, dtype=np.float32).reshape(1, -1)
        results, stats = self.search_batch_with_stats(q, k, probes, rerank_factor)
        return results[0], stats[0]

    def search_batch_with_stats(self, queries: np.ndarray, k: int, probes: int, rerank_factor: int):
        """search_with_stats for every row: ([results per query], [SearchStats per query])."""
        ids, dist, count, leaves, lcount, lvl, lvlc, scanned, reranked = self.dev.search_batch_with_stats(
            queries, k, probes, rerank_factor)
        total_leaves = int(self.flat.is_leaf().sum())
        results, stats = [], []
        for i in range(ids.shape[0]):
            results.append([(int(ids[i, j]), float(dist[i, j])) for j in range(int(count[i]))])
            stats.append(SearchStats(
                total_leaves=total_leaves, leaves_searched=int(lcount[i]),
                leaves_searched_ids=[int(v) for v in leaves[i, :lcount[i]]], total_vectors=self.flat.n_vectors,
                vectors_scanned_binary=int(scanned[i]), vectors_reranked_full=int(reranked[i]),
                tree_depth=max(self.dev.levels - 1, 0), probes_per_level=[int(v) for v in lvl[i, :lvlc[i]]]))
        return results, stats

    def len(self) -> int:
        return len(self)

    def is_empty(self) -> bool:
        return len(self) == 0


class ClusteredIndexWithRaBitQ(ClusteredIndex):
    """ClusteredIndexWithRaBitQ (hierarchical_with_rabitq.rs): same tree, RaBitQ phase-1 codes."""

    code_kind = "rabitq"
