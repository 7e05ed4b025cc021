"""ctypes wrapper of oracle/libcuddle_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (vectordb_amd) never does. See the header of cuddle_oracle.c for what the oracle is and
how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from vectordb_amd import _ffi
from vectordb_amd.flat_index import FlatIndex

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcuddle_oracle.so")

ORC_OK, ORC_ERR_BAD_ARG, ORC_ERR_BAD_DIM, ORC_ERR_OOM, ORC_ERR_NAN = 0, 1, 2, 5, 100


class Trace(C.Structure):
    """struct orc_trace."""

    _fields_ = [
        ("leaves", C.c_void_p), ("leaves_cap", C.c_uint32), ("n_leaves", C.c_uint32),
        ("frontier_sizes", C.c_void_p), ("levels_cap", C.c_uint32), ("n_levels", C.c_uint32),
        ("cand_pos", C.c_void_p), ("cand_key", C.c_void_p), ("cand_cap", C.c_uint64), ("n_cand", C.c_uint64),
        ("sel_pos", C.c_void_p), ("sel_key", C.c_void_p), ("sel_dist", C.c_void_p), ("sel_cap", C.c_uint64),
        ("n_sel", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "cuddle_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
        sig = {
            "orc_dot": (f32, [vp, vp, C.c_size_t]),
            "orc_l2": (f32, [vp, vp, C.c_size_t]),
            "orc_l2_batch_rows": (None, [vp, vp, C.c_size_t, C.c_size_t, vp]),
            "orc_distance": (f32, [vp, vp, C.c_size_t, C.c_int]),
            "orc_distance_pairs": (None, [vp, vp, u64, u32, C.c_int, vp]),
            "orc_kmeans_assign": (None, [vp, u64, vp, u32, u32, C.c_int, vp, vp]),
            "orc_kmeans_update": (None, [vp, u64, u32, vp, u32, vp, vp]),
            "orc_kmeanspp_init": (None, [vp, u64, u32, u32, C.c_int, u64, vp, vp, vp]),
            "orc_binary_quantize": (None, [vp, u64, u32, f32, vp]),
            "orc_binary_threshold": (f32, [vp, u64, u32]),
            "orc_hamming": (u32, [vp, vp, u32]),
            "orc_hamming_pairs": (None, [vp, vp, u64, u32, vp]),
            "orc_rabitq_quantize": (None, [vp, u32, vp, u64, vp, vp, vp]),
            "orc_rabitq_rotate": (None, [vp, u32, vp, vp]),
            "orc_rabitq_estimate_fast": (f32, [vp, f32, f32, u32, f32, vp]),
            "orc_rabitq_query_prep": (f32, [vp, u32, vp, vp]),
            "orc_rabitq_estimate_positions": (None, [C.POINTER(_ffi.IndexDesc), vp, vp, u64, vp]),
            "orc_search": (C.c_int, [C.POINTER(_ffi.IndexDesc), vp, u32, u32, u32, u32, u32, vp, vp, vp, C.POINTER(Trace)]),
            "orc_search_leaves": (C.c_int, [C.POINTER(_ffi.IndexDesc), vp, vp, u32, u32, u32, u32, vp, vp, vp, C.POINTER(Trace)]),
            "orc_search_batch": (C.c_int, [C.POINTER(_ffi.IndexDesc), vp, u32, u32, u32, u32, u32, u32, vp, vp, vp, C.c_int]),
            "orc_search_sequential": (C.c_int, [C.POINTER(_ffi.IndexDesc), vp, u32, u32, u32, u32, u32, u32, vp, vp, vp, C.c_int]),
            "orc_num_threads": (C.c_int, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def _p(a):
    return _ffi.ptr(a)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- leaf routines -----------------------------------------------------------------------------
def dot(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot(_p(a), _p(b), a.size))


def l2(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_l2(_p(a), _p(b), a.size))


def l2_batch(query, targets) -> np.ndarray:
    q, t = _f32(query), _f32(targets)
    out = np.zeros(t.shape[0], dtype=np.float32)
    lib().orc_l2_batch_rows(_p(q), _p(t), t.shape[0], t.shape[1] if t.ndim == 2 else 0, _p(out))
    return out


def distance(a, b, metric: int) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_distance(_p(a), _p(b), a.size, metric))


def distance_pairs(a, b, metric: int) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    out = np.zeros(a.shape[0], dtype=np.float32)
    lib().orc_distance_pairs(_p(a), _p(b), a.shape[0], a.shape[1], metric, _p(out))
    return out


def kmeans_assign(vectors, centroids, metric: int):
    """KMeans::assign (kmeans.rs:136-160): (index of the nearest centroid, its distance) per vector."""
    v, c = _f32(vectors), _f32(centroids)
    assign = np.zeros(v.shape[0], dtype=np.uint32)
    dist = np.zeros(v.shape[0], dtype=np.float32)
    lib().orc_kmeans_assign(_p(v), v.shape[0], _p(c), c.shape[0], v.shape[1], metric, _p(assign), _p(dist))
    return assign, dist


def kmeans_update(vectors, assign, k: int, centroids=None):
    """KMeans::update_centroids (kmeans.rs:163-215): (centroids [k, d], counts [k]); rows of empty
    clusters keep the value `centroids` had (zeros when not given)."""
    v = _f32(vectors)
    a = np.ascontiguousarray(assign, dtype=np.uint32)
    c = np.zeros((k, v.shape[1]), dtype=np.float32) if centroids is None else _f32(centroids).copy()
    counts = np.zeros(k, dtype=np.uint32)
    lib().orc_kmeans_update(_p(v), v.shape[0], v.shape[1], _p(a), k, _p(c), _p(counts))
    return c, counts


def kmeanspp_init(vectors, k: int, metric: int, first: int, uniforms):
    """KMeans::init_plusplus (kmeans.rs:36-88) with the random draws given: (centroids [k, d], chosen [k])."""
    v = _f32(vectors)
    u = _f32(uniforms)
    assert u.size >= k - 1
    c = np.zeros((k, v.shape[1]), dtype=np.float32)
    chosen = np.zeros(k, dtype=np.uint64)
    lib().orc_kmeanspp_init(_p(v), v.shape[0], v.shape[1], k, metric, first, _p(u), _p(c), _p(chosen))
    return c, chosen


def binary_quantize(v, threshold: float) -> np.ndarray:
    v = _f32(v)
    if v.ndim == 1:
        v = v[None, :]
    out = np.zeros((v.shape[0], (v.shape[1] + 63) // 64), dtype=np.uint64)
    lib().orc_binary_quantize(_p(v), v.shape[0], v.shape[1], threshold, _p(out))
    return out


def binary_threshold(v) -> float:
    v = _f32(v)
    return float(lib().orc_binary_threshold(_p(v), v.shape[0], v.shape[1]))


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(lib().orc_hamming(_p(a), _p(b), a.size))


def hamming_pairs(a, b) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(a.shape[0], dtype=np.uint32)
    lib().orc_hamming_pairs(_p(a), _p(b), a.shape[0], a.shape[1], _p(out))
    return out


def rabitq_quantize(rot_t, v):
    rot_t, v = _f32(rot_t), _f32(v)
    if v.ndim == 1:
        v = v[None, :]
    n, d = v.shape
    codes = np.zeros((n, (d + 31) // 32), dtype=np.uint32)
    norms = np.zeros(n, dtype=np.float32)
    ipoo = np.zeros(n, dtype=np.float32)
    lib().orc_rabitq_quantize(_p(rot_t), d, _p(v), n, _p(codes), _p(norms), _p(ipoo))
    return codes, norms, ipoo


def rabitq_query_prep(rot_t, q):
    rot_t, q = _f32(rot_t), _f32(q)
    rq = np.zeros(q.size, dtype=np.float32)
    qnorm = float(lib().orc_rabitq_query_prep(_p(rot_t), q.size, _p(q), _p(rq)))
    return qnorm, rq


def rabitq_estimate_fast(code, norm: float, ipoo: float, dim: int, qnorm: float, rq) -> float:
    code = np.ascontiguousarray(code, dtype=np.uint32)
    rq = _f32(rq)
    return float(lib().orc_rabitq_estimate_fast(_p(code), norm, ipoo, dim, qnorm, _p(rq)))


# ---- search ------------------------------------------------------------------------------------
class OracleIndex:
    """The oracle's view of a FlatIndex (it reads the same cuddle_index_desc the product uploads)."""

    def __init__(self, flat: FlatIndex):
        self.flat = flat.normalised()
        self.desc = self.flat.desc()

    def search(self, query, k: int, probes: int, rerank_factor: int, flags: int = 0, trace: bool = False):
        f = self.flat
        q = _f32(query)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        cnt = C.c_uint32(0)
        tr = None
        keep = {}
        if trace:
            tr = Trace()
            n_nodes, n_vec = f.n_nodes, max(f.n_vectors, 1)
            keep["leaves"] = np.zeros(n_nodes, dtype=np.uint32)
            keep["fs"] = np.zeros(80, dtype=np.uint32)
            keep["cpos"] = np.zeros(n_vec, dtype=np.uint64)
            keep["ckey"] = np.zeros(n_vec, dtype=np.uint32)
            keep["spos"] = np.zeros(n_vec, dtype=np.uint64)
            keep["skey"] = np.zeros(n_vec, dtype=np.uint32)
            keep["sdist"] = np.zeros(n_vec, dtype=np.float32)
            tr.leaves, tr.leaves_cap = keep["leaves"].ctypes.data, n_nodes
            tr.frontier_sizes, tr.levels_cap = keep["fs"].ctypes.data, 80
            tr.cand_pos, tr.cand_key, tr.cand_cap = keep["cpos"].ctypes.data, keep["ckey"].ctypes.data, n_vec
            tr.sel_pos, tr.sel_key, tr.sel_dist = keep["spos"].ctypes.data, keep["skey"].ctypes.data, keep["sdist"].ctypes.data
            tr.sel_cap = n_vec
        rc = lib().orc_search(C.byref(self.desc), _p(q), q.size, k, probes, rerank_factor, flags, _p(ids), _p(dist),
                              C.byref(cnt), C.byref(tr) if tr is not None else None)
        out = {"rc": rc, "ids": ids[:k], "dist": dist[:k], "count": int(cnt.value)}
        if trace and rc in (ORC_OK, ORC_ERR_NAN):
            out["leaves"] = keep["leaves"][:tr.n_leaves].copy()
            out["frontier_sizes"] = keep["fs"][:tr.n_levels].copy()
            out["cand_pos"] = keep["cpos"][:tr.n_cand].copy()
            out["cand_key"] = keep["ckey"][:tr.n_cand].copy()
            out["sel_pos"] = keep["spos"][:tr.n_sel].copy()
            out["sel_key"] = keep["skey"][:tr.n_sel].copy()
            out["sel_dist"] = keep["sdist"][:tr.n_sel].copy()
        return out

    def search_batch(self, queries, k: int, probes: int, rerank_factor: int, flags: int = 0, n_threads: int = 0):
        q = _f32(queries)
        nq, dim = q.shape
        ids = np.zeros((nq, k), dtype=np.uint64)
        dist = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        rc = lib().orc_search_batch(C.byref(self.desc), _p(q), nq, dim, k, probes, rerank_factor, flags, _p(ids),
                                    _p(dist), _p(cnt), n_threads)
        return rc, ids, dist, cnt

    def search_sequential(self, queries, k: int, probes: int, rerank_factor: int, flags: int = 0, n_threads: int = 0):
        """One search() after another, each using the thread pool inside (the reference's own shape)."""
        q = _f32(queries)
        nq, dim = q.shape
        ids = np.zeros((nq, k), dtype=np.uint64)
        dist = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        rc = lib().orc_search_sequential(C.byref(self.desc), _p(q), nq, dim, k, probes, rerank_factor, flags, _p(ids),
                                         _p(dist), _p(cnt), n_threads)
        return rc, ids, dist, cnt


def num_threads() -> int:
    return int(lib().orc_num_threads())
