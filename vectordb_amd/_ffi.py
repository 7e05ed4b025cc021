"""ctypes binding of libcuddle_hip.so (declared in include/cuddle_hip.h).

The product path has no CPU fallback: importing this module without the built library raises,
and every compute call on a machine without a gfx950 device returns CUDDLE_ERR_NO_DEVICE, which
`check()` turns into a CuddleError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUDDLE_HIP_LIB selects another build of the same library (kernel tuning experiments).
LIB_PATH = os.environ.get("CUDDLE_HIP_LIB") or os.path.join(_HERE, "libcuddle_hip.so")

ABI_VERSION = 2

OK, ERR_BAD_ARG, ERR_BAD_DIM, ERR_HIP, ERR_NO_DEVICE, ERR_OOM, ERR_BAD_INDEX, ERR_UNSUPPORTED, ERR_RCCL = range(9)
STATUS_NAMES = {
    OK: "CUDDLE_OK",
    ERR_BAD_ARG: "CUDDLE_ERR_BAD_ARG",
    ERR_BAD_DIM: "CUDDLE_ERR_BAD_DIM",
    ERR_HIP: "CUDDLE_ERR_HIP",
    ERR_NO_DEVICE: "CUDDLE_ERR_NO_DEVICE",
    ERR_OOM: "CUDDLE_ERR_OOM",
    ERR_BAD_INDEX: "CUDDLE_ERR_BAD_INDEX",
    ERR_UNSUPPORTED: "CUDDLE_ERR_UNSUPPORTED",
    ERR_RCCL: "CUDDLE_ERR_RCCL",
}

METRIC_L2, METRIC_COSINE, METRIC_DOT = 0, 1, 2
CODE_BINARY, CODE_RABITQ = 0, 1
SEARCH_STATS_VARIANT = 0x1
SEARCH_TIME_STAGES = 0x2
SEARCH_ROUTE_EXACT_ONLY = 0x4
SEARCH_ROUTE_GROUPED = 0x8
SEARCH_ROUTE_SPLIT_FILTER = 0x10
DESC_DEVICE_ARRAYS = 0x1
DESC_ADOPT_VECTORS = 0x2
STAGE_NAMES = ("quantize", "route", "scan", "rerank", "topk", "scan_prep")


class CuddleError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class IndexDesc(C.Structure):
    """struct cuddle_index_desc."""

    _fields_ = [
        ("abi_version", C.c_uint32),
        ("dim", C.c_uint32),
        ("metric", C.c_uint32),
        ("code_kind", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("n_roots", C.c_uint32),
        ("n_vectors", C.c_uint64),
        ("centroids", C.c_void_p),
        ("child_offsets", C.c_void_p),
        ("children", C.c_void_p),
        ("roots", C.c_void_p),
        ("vec_offsets", C.c_void_p),
        ("ids", C.c_void_p),
        ("vectors", C.c_void_p),
        ("codes", C.c_void_p),
        ("threshold", C.c_float),
        ("flags", C.c_uint32),
        ("rabitq_rot_t", C.c_void_p),
        ("rabitq_codes", C.c_void_p),
        ("rabitq_norm", C.c_void_p),
        ("rabitq_ip_oo", C.c_void_p),
    ]


class Stats(C.Structure):
    """struct cuddle_stats."""

    _fields_ = [
        ("leaves_searched", C.c_uint64),
        ("vectors_scanned", C.c_uint64),
        ("vectors_reranked", C.c_uint64),
        ("max_candidates", C.c_uint32),
        ("max_leaves", C.c_uint32),
        ("tree_levels", C.c_uint32),
        ("route_fallbacks", C.c_uint32),
        ("rabitq_exact_estimates", C.c_uint64),
        ("rabitq_exact_all_queries", C.c_uint64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# Every symbol include/cuddle_hip.h declares: name -> (restype, argtypes)
_u32, _u64, _f32p, _vp, _int = C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int
SIGNATURES = {
    "cuddle_abi_version": (_u32, []),
    "cuddle_last_error": (C.c_char_p, []),
    "cuddle_device_count": (_int, []),
    "cuddle_index_create": (_int, [C.POINTER(IndexDesc), _int, C.POINTER(_vp)]),
    "cuddle_index_destroy": (None, [_vp]),
    "cuddle_index_reserve": (_int, [_vp, _u32, _u32, _u32, _u32]),
    "cuddle_search_batch": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(Stats)]),
    "cuddle_search_batch_device": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "cuddle_search_batch_device_noalloc": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "cuddle_search_batch_with_stats": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32,
                                              _vp, _vp, _vp]),
    "cuddle_last_stats": (_int, [_vp, C.POINTER(Stats)]),
    "cuddle_stage_times": (_int, [_vp, _vp, _vp]),
    "cuddle_distance_pairs": (_int, [_vp, _vp, _u64, _u32, _u32, _vp, _int]),
    "cuddle_binary_quantize": (_int, [_vp, _u64, _u32, C.c_float, _vp, _int]),
    "cuddle_hamming_pairs": (_int, [_vp, _vp, _u64, _u32, _vp, _int]),
    "cuddle_binary_threshold": (C.c_float, [_vp, _u64, _u32]),
    "cuddle_route_batch": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _u32, _vp]),
    "cuddle_route_filter_error": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "cuddle_scan_mode": (_int, [_vp, _u32, _u32, _vp]),
    "cuddle_scan_read_ceiling": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "cuddle_index_leaf_stride": (_int, [_vp, _u32, _vp]),
    "cuddle_route_batch_device": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp]),
    "cuddle_shard_scan_leaves": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp, _vp]),
    "cuddle_kmeans_assign": (_int, [_vp, _u64, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _int, _vp]),
    "cuddle_kmeans_update": (_int, [_vp, _u64, _u32, _vp, _u32, _vp, _vp, _int, _vp]),
    "cuddle_kmeanspp_init": (_int, [_vp, _u64, _u32, _u32, _u32, _u64, _vp, _vp, _vp, _int, _vp]),
    "cuddle_binary_encode_device": (_int, [_vp, _u64, _u32, C.c_float, _vp, _int, _vp]),
    "cuddle_rabitq_encode_device": (_int, [_vp, _u32, _vp, _u64, _vp, _vp, _vp, _int, _vp]),
    "cuddle_scan_batch": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _vp, _vp, _vp, _u32, _vp]),
    "cuddle_search_leaves_batch": (
        _int,
        [_vp, _vp, _u32, _u32, _vp, _u32, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    ),
    "cuddle_index_dim": (_u32, [_vp]),
    "cuddle_index_levels": (_u32, [_vp]),
    "cuddle_index_max_children": (_u32, [_vp]),
    "cuddle_index_max_leaf_size": (_u32, [_vp]),
    "cuddle_index_device_bytes": (_u64, [_vp]),
    "cuddle_merge_topk": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "cuddle_shard_scan": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, C.POINTER(_u32), _vp]),
    "cuddle_shard_ties": (_int, [_vp, _vp, _vp, _vp]),
    "cuddle_shard_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cuddle_rerank_k": (_int, [_u32, _u32, _u32, _vp]),
    "cuddle_shard_scan_rabitq": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp]),
    "cuddle_shard_finish_rabitq": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "cuddle_merge_topk_ordered": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "cuddle_comm_unique_id": (_int, [_vp]),
    "cuddle_comm_init": (_int, [_int, _int, _vp, _int, C.POINTER(_vp)]),
    "cuddle_comm_destroy": (None, [_vp]),
    "cuddle_comm_rank": (_int, [_vp]),
    "cuddle_comm_world": (_int, [_vp]),
    "cuddle_sharded_search_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "cuddle_merge_topk_ordered_device": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u64, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libcuddle_hip.so (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C vectordb_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback."
            )
        # torch ships its own copy of the HIP runtime under the system's soname; the copy loaded first serves the whole
        # process, and torch finds the GPU only through its own - so where torch is installed it goes first
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.cuddle_abi_version() != ABI_VERSION:
            raise ImportError("libcuddle_hip.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def check(status: int) -> None:
    if status != OK:
        raise CuddleError(status, lib().cuddle_last_error().decode("utf-8", "replace"))


def ptr(a) -> C.c_void_p:
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return C.c_void_p(None)
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need a C-contiguous numpy array"
    return C.c_void_p(a.ctypes.data)
