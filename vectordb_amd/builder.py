"""Index construction: produces the FlatIndex the query path runs on.

Follows the reference's build rules (/root/reference/src/index/hierarchical.rs:188-321,
src/clustering/kmeans.rs:36-226) but is NOT a parity target: the reference seeds k-means and the
RaBitQ rotation from `thread_rng()`, so no two of its own builds agree; the index is an *input*
of the query path. Everything here is seeded and deterministic, runs on any torch device (the
distance matrices are plain matmuls on the GPU), and is used by tests, the benchmark and
`ClusteredIndex.build`.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

from . import _ffi
from .flat_index import FlatIndex, pack_binary_codes

_METRICS = {"l2": _ffi.METRIC_L2, "cosine": _ffi.METRIC_COSINE, "dot": _ffi.METRIC_DOT,
            _ffi.METRIC_L2: _ffi.METRIC_L2, _ffi.METRIC_COSINE: _ffi.METRIC_COSINE, _ffi.METRIC_DOT: _ffi.METRIC_DOT}


def metric_id(metric) -> int:
    if isinstance(metric, str):
        metric = metric.lower()
    return _METRICS[metric]


def _pairwise(x: torch.Tensor, c: torch.Tensor, metric: int) -> torch.Tensor:
    """distance() of every row of x to every row of c (distance/mod.rs:23-34), matmul form."""
    dots = x @ c.t()
    if metric == _ffi.METRIC_DOT:
        return -dots
    if metric == _ffi.METRIC_COSINE:
        xn = x.norm(dim=1, keepdim=True).clamp_min(1e-30)
        cn = c.norm(dim=1, keepdim=True).clamp_min(1e-30)
        return 1.0 - dots / (xn * cn.t())
    return (x * x).sum(1, keepdim=True) + (c * c).sum(1)[None, :] - 2.0 * dots


def kmeans_assign(x: torch.Tensor, c: torch.Tensor, metric, stream=None):
    """KMeans::assign (kmeans.rs:136-160) on the GPU through the C ABI (cuddle_kmeans_assign): for
    CUDA tensors x [n, d] and c [k, d] returns (assignment int32 [n], distance f32 [n], number of
    vectors that needed the all-centroids exact comparison). Exact: distance() in the reference's
    arithmetic, first of equal minima."""
    if not (x.is_cuda and c.is_cuda):
        raise ValueError("kmeans_assign runs on the GPU: pass CUDA tensors")
    x = x.contiguous().float()
    c = c.contiguous().float()
    if x.shape[1] != c.shape[1]:
        raise ValueError("dimension mismatch")
    assign = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    dist = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    fb = np.zeros(1, dtype=np.uint64)
    st = stream if stream is not None else torch.cuda.current_stream(x.device)
    _ffi.check(_ffi.lib().cuddle_kmeans_assign(x.data_ptr(), x.shape[0], c.data_ptr(), c.shape[0], x.shape[1],
                                               metric_id(metric), assign.data_ptr(), dist.data_ptr(), _ffi.ptr(fb),
                                               x.device.index or 0, st.cuda_stream))
    return assign, dist, int(fb[0])


def _stream(t: torch.Tensor, stream):
    return (stream if stream is not None else torch.cuda.current_stream(t.device)).cuda_stream


def kmeans_update(x: torch.Tensor, assign: torch.Tensor, k: int, centroids: Optional[torch.Tensor] = None, stream=None):
    """KMeans::update_centroids (kmeans.rs:163-215) on the GPU (cuddle_kmeans_update): (centroids [k, d]
    f32, counts [k] int32). Rows of empty clusters keep what `centroids` held (zeros when not given):
    re-seeding them (:217-223) is a random draw and stays with the caller."""
    if not x.is_cuda:
        raise ValueError("kmeans_update runs on the GPU: pass CUDA tensors")
    x = x.contiguous().float()
    a = assign.to(torch.int32).contiguous()
    cent = torch.zeros((k, x.shape[1]), dtype=torch.float32, device=x.device) if centroids is None else centroids.contiguous().float().clone()
    counts = torch.empty(k, dtype=torch.int32, device=x.device)
    _ffi.check(_ffi.lib().cuddle_kmeans_update(x.data_ptr(), x.shape[0], x.shape[1], a.data_ptr(), k, cent.data_ptr(),
                                               counts.data_ptr(), x.device.index or 0, _stream(x, stream)))
    return cent, counts


def kmeanspp_init(x: torch.Tensor, k: int, metric, first: int, uniforms, stream=None):
    """KMeans::init_plusplus (kmeans.rs:36-88) with the random draws given (cuddle_kmeanspp_init):
    (centroids [k, d] CUDA f32, chosen [k] uint64 numpy)."""
    if not x.is_cuda:
        raise ValueError("kmeanspp_init runs on the GPU: pass a CUDA tensor")
    x = x.contiguous().float()
    u = np.ascontiguousarray(uniforms, dtype=np.float32)
    if u.size < k - 1:
        raise ValueError("need k - 1 uniform draws")
    cent = torch.empty((k, x.shape[1]), dtype=torch.float32, device=x.device)
    chosen = np.zeros(k, dtype=np.uint64)
    _ffi.check(_ffi.lib().cuddle_kmeanspp_init(x.data_ptr(), x.shape[0], x.shape[1], k, metric_id(metric), int(first), _ffi.ptr(u),
                                               cent.data_ptr(), _ffi.ptr(chosen), x.device.index or 0, _stream(x, stream)))
    return cent, chosen


def binary_encode_device(x: torch.Tensor, threshold: float, stream=None) -> torch.Tensor:
    """BinaryQuantizer::quantize (binary.rs:144-159) for CUDA rows: codes as int64 [n, ceil(d/64)] (the u64 bit patterns)."""
    if not x.is_cuda:
        raise ValueError("binary_encode_device runs on the GPU: pass a CUDA tensor")
    x = x.contiguous().float()
    codes = torch.empty((x.shape[0], (x.shape[1] + 63) // 64), dtype=torch.int64, device=x.device)
    _ffi.check(_ffi.lib().cuddle_binary_encode_device(x.data_ptr(), x.shape[0], x.shape[1], float(threshold), codes.data_ptr(),
                                                      x.device.index or 0, _stream(x, stream)))
    return codes


def rabitq_encode_device(x: torch.Tensor, rot_t: torch.Tensor, stream=None):
    """RaBitQQuantizer::quantize_batch (rabitq.rs:114-179) for CUDA rows, sums in the reference's order:
    (codes int32 [n, ceil(d/32)] holding the u32 bit patterns, norms f32 [n], ip_oo f32 [n])."""
    if not (x.is_cuda and rot_t.is_cuda):
        raise ValueError("rabitq_encode_device runs on the GPU: pass CUDA tensors")
    x = x.contiguous().float()
    r = rot_t.contiguous().float()
    n, d = x.shape
    codes = torch.empty((n, (d + 31) // 32), dtype=torch.int32, device=x.device)
    norms = torch.empty(n, dtype=torch.float32, device=x.device)
    ipoo = torch.empty(n, dtype=torch.float32, device=x.device)
    _ffi.check(_ffi.lib().cuddle_rabitq_encode_device(r.data_ptr(), d, x.data_ptr(), n, codes.data_ptr(), norms.data_ptr(),
                                                      ipoo.data_ptr(), x.device.index or 0, _stream(x, stream)))
    return codes, norms, ipoo


def _assign(x: torch.Tensor, c: torch.Tensor, metric: int, chunk: int = 262144) -> torch.Tensor:
    if x.is_cuda:  # native exact assignment; a missing library is an error, not a fallback
        return kmeans_assign(x, c, metric)[0].long()
    out = torch.empty(x.shape[0], dtype=torch.long, device=x.device)
    for s in range(0, x.shape[0], chunk):
        out[s:s + chunk] = _pairwise(x[s:s + chunk], c, metric).argmin(dim=1)
    return out


def kmeans_fit(x: torch.Tensor, k: int, metric: int, max_iterations: int, gen: torch.Generator):
    """KMeans::fit (kmeans.rs:93-133): k-means++ seeding that samples proportionally to the
    *square* of distance() (which for L2 is already squared, :67-72), at most min(max_iterations,
    12) Lloyd iterations, stop when < 2 % of the points change cluster; empty clusters are
    re-seeded with a random point (:217-223). Returns (centroids [k,d], assignment [n])."""
    n = x.shape[0]
    dev = x.device
    if x.is_cuda:
        return _kmeans_fit_native(x, k, metric, max_iterations, gen)
    first = int(torch.randint(0, n, (1,), generator=gen).item())
    cent = torch.empty((k, x.shape[1]), dtype=x.dtype, device=dev)
    cent[0] = x[first]
    mind = _pairwise(x, cent[0:1], metric)[:, 0]
    for j in range(1, k):
        w = (mind * mind).clamp_min(0).double()
        total = float(w.sum().item())
        if not math.isfinite(total) or total <= 0.0:
            idx = int(torch.randint(0, n, (1,), generator=gen).item())
        else:
            target = float(torch.rand((1,), generator=gen).item()) * total
            idx = int(torch.searchsorted(torch.cumsum(w, 0), torch.tensor([target], dtype=torch.float64, device=dev)).item())
            idx = min(idx, n - 1)
        cent[j] = x[idx]
        mind = torch.minimum(mind, _pairwise(x, cent[j:j + 1], metric)[:, 0])
    assign = _assign(x, cent, metric)
    prev = assign.clone()
    for _ in range(min(max_iterations, 12)):
        sums = torch.zeros_like(cent)
        sums.index_add_(0, assign, x)
        counts = torch.bincount(assign, minlength=k)
        nonempty = counts > 0
        cent = torch.where(nonempty[:, None], sums / counts.clamp_min(1)[:, None].to(x.dtype), cent)
        for e in torch.nonzero(~nonempty).flatten().tolist():
            cent[e] = x[int(torch.randint(0, n, (1,), generator=gen).item())]
        assign = _assign(x, cent, metric)
        changed = int((assign != prev).sum().item())
        if changed / n < 0.02:
            break
        prev = assign.clone()
    return cent, assign


def _kmeans_fit_native(x: torch.Tensor, k: int, metric: int, max_iterations: int, gen: torch.Generator):
    """kmeans_fit for CUDA tensors through the library's build entry points, each of which follows the
    reference's arithmetic: seeding (cuddle_kmeanspp_init), assignment (cuddle_kmeans_assign), centroid
    update (cuddle_kmeans_update). Only the random draws are made here."""
    n = x.shape[0]
    first = int(torch.randint(0, n, (1,), generator=gen).item())
    u = torch.rand((max(k - 1, 1),), generator=gen).numpy()
    cent, _ = kmeanspp_init(x, k, metric, first, u)
    assign = kmeans_assign(x, cent, metric)[0]
    prev = assign.clone()
    for _ in range(min(max_iterations, 12)):
        cent, counts = kmeans_update(x, assign, k, cent)
        for e in torch.nonzero(counts == 0).flatten().tolist():  # kmeans.rs:217-223
            cent[e] = x[int(torch.randint(0, n, (1,), generator=gen).item())]
        assign = kmeans_assign(x, cent, metric)[0]
        changed = int((assign != prev).sum().item())
        if changed / n < 0.02:
            break
        prev = assign.clone()
    return cent, assign.long()


class _Tree:
    def __init__(self):
        self.centroid: List[torch.Tensor] = []
        self.children: List[List[int]] = []
        self.vec: List[Optional[torch.Tensor]] = []

    def add(self, centroid, children, vec) -> int:
        self.centroid.append(centroid)
        self.children.append(children)
        self.vec.append(vec)
        return len(self.centroid) - 1


def _build_recursive(tree: _Tree, x: torch.Tensor, idx: torch.Tensor, depth: int, target_leaf: int, branching: int,
                     metric: int, max_iterations: int, gen: torch.Generator) -> List[int]:
    """build_recursive (hierarchical.rs:188-321)."""
    n = idx.numel()
    if n == 0:
        return []
    if n <= 2 * target_leaf or depth >= 10:  # :210
        sub = x[idx]
        return [tree.add(sub.sum(0) / n, [], idx)]  # leaf centroid = mean (:509-531)
    k = min(max(math.ceil(n / target_leaf), 2), branching, n)  # :221-225
    sub = x[idx]
    cent, assign = kmeans_fit(sub, k, metric, max_iterations, gen)
    clusters = []
    for c in range(k):
        m = torch.nonzero(assign == c).flatten()
        if m.numel():
            clusters.append([c, idx[m]])
    min_size = max(target_leaf * 3 // 10, 10)  # :254
    i = 0
    while i < len(clusters):
        if clusters[i][1].numel() < min_size and len(clusters) > 1:
            small = clusters.pop(i)
            others = torch.stack([cent[c] for c, _ in clusters])
            best = int(_pairwise(cent[small[0]][None, :], others, metric)[0].argmin().item())
            clusters[best][1] = torch.cat([clusters[best][1], small[1]])
        else:
            i += 1
    out = []
    for c, members in clusters:
        kids = _build_recursive(tree, x, members, depth + 1, target_leaf, branching, metric, max_iterations, gen)
        if len(kids) == 1:  # :299-300
            out.append(kids[0])
        else:
            out.append(tree.add(cent[c].clone(), kids, None))  # internal centroid = k-means centroid (:307-313)
    return out


def make_rotation(dim: int, seed: int) -> np.ndarray:
    """A seeded random orthogonal matrix, returned transposed like rotation_matrix_t (rabitq.rs:20,
    :60-105 builds one by Gram-Schmidt on Gaussian rows; any orthogonal matrix is a valid input)."""
    rng = np.random.default_rng(seed)
    q, r = np.linalg.qr(rng.standard_normal((dim, dim)))
    q = q * np.sign(np.diag(r))[None, :]
    return np.ascontiguousarray(q.T.astype(np.float32))


def rabitq_encode(vectors: torch.Tensor, rot_t: torch.Tensor, chunk: int = 131072):
    """RaBitQQuantizer::quantize (rabitq.rs:114-171) for a matrix, matmul form: norm, sign code of
    the rotated unit vector (bit i of word i//32), <o_bar, o>. Zero vectors: code 0, norm 0, 1.0."""
    n, d = vectors.shape
    words = (d + 31) // 32
    codes = np.zeros((n, words), dtype=np.uint32)
    norms = np.zeros(n, dtype=np.float32)
    ipoo = np.ones(n, dtype=np.float32)
    for s in range(0, n, chunk):
        v = vectors[s:s + chunk].float()
        nr = v.norm(dim=1)
        zero = nr < 1e-10
        rot = (v / nr.clamp_min(1e-30)[:, None]) @ rot_t.t()
        pos = rot >= 0
        ip = rot.abs().sum(1)
        bits = torch.zeros((v.shape[0], words * 32), dtype=torch.uint8, device=v.device)
        bits[:, :d] = pos
        bits[zero] = 0
        b = np.packbits(bits.cpu().numpy().reshape(-1, words, 4, 8), axis=-1, bitorder="little")
        codes[s:s + chunk] = np.ascontiguousarray(b.reshape(-1, words, 4)).view("<u4").reshape(-1, words)
        norms[s:s + chunk] = torch.where(zero, torch.zeros_like(nr), nr).cpu().numpy()
        ipoo[s:s + chunk] = torch.where(zero, torch.ones_like(ip), ip).cpu().numpy()
    return codes, norms, ipoo


def pack_binary_codes_torch(x: torch.Tensor, threshold: float, chunk: int = 65536) -> np.ndarray:
    """pack_binary_codes on a torch device (same bit layout), returned as uint64 numpy."""
    n, d = x.shape
    words = (d + 63) // 64
    out = np.empty((n, words), dtype=np.uint64)
    sh = torch.arange(32, device=x.device, dtype=torch.int64)
    for s in range(0, n, chunk):
        b = torch.zeros((min(chunk, n - s), words * 64), dtype=torch.int64, device=x.device)
        b[:, :d] = (x[s:s + chunk] > threshold).to(torch.int64)
        halves = (b.view(-1, words, 2, 32) << sh).sum(-1)          # [m, words, 2] 32-bit halves
        w64 = halves[..., 0] | (halves[..., 1] << 32)
        out[s:s + chunk] = w64.cpu().numpy().view(np.uint64)
    return out


def binary_threshold(vectors_f32: np.ndarray) -> float:
    """BinaryQuantizer::from_vectors (binary.rs:122-141), computed by the library's host routine."""
    v = np.ascontiguousarray(vectors_f32, dtype=np.float32)
    return float(_ffi.lib().cuddle_binary_threshold(_ffi.ptr(v), v.shape[0], v.shape[1]))


def build_index(vectors, branching_factor: int = 100, target_leaf_size: int = 100, metric="l2",
                max_iterations: int = 20, seed: int = 0, code_kind: str = "binary", device: Optional[str] = None,
                threshold: Optional[float] = None) -> FlatIndex:
    """ClusteredIndex::build / ClusteredIndexWithRaBitQ::build (hierarchical.rs:87-174,
    hierarchical_with_rabitq.rs:84-170), returning the flat form. `vectors`: [N, D] numpy or torch,
    in the caller's id order (ids are 0..N-1)."""
    assert branching_factor >= 2 and target_leaf_size >= 10  # :96-97
    m = metric_id(metric)
    x = torch.as_tensor(vectors)
    assert x.dim() == 2 and x.shape[0] > 0
    dev = torch.device(device) if device is not None else x.device
    x = x.to(device=dev, dtype=torch.float32)
    n, d = x.shape
    gen = torch.Generator(device="cpu")  # scalar draws only; the heavy math runs on `dev`
    gen.manual_seed(seed)
    tree = _Tree()
    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        roots = _build_recursive(tree, x, torch.arange(n, device=dev), 0, target_leaf_size, branching_factor, m,
                                 max_iterations, gen)
    finally:
        sys.setrecursionlimit(old)
    n_nodes = len(tree.centroid)
    centroids = torch.stack(tree.centroid).cpu().numpy().astype(np.float32)
    child_offsets = np.zeros(n_nodes + 1, dtype=np.uint32)
    vec_offsets = np.zeros(n_nodes + 1, dtype=np.uint64)
    children: List[int] = []
    order = []
    for i in range(n_nodes):
        children.extend(tree.children[i])
        child_offsets[i + 1] = len(children)
        cnt = 0 if tree.vec[i] is None else int(tree.vec[i].numel())
        vec_offsets[i + 1] = vec_offsets[i] + np.uint64(cnt)
        if cnt:
            order.append(tree.vec[i])
    ids_t = torch.cat(order) if order else torch.zeros(0, dtype=torch.long, device=dev)
    vec_lc = x[ids_t].cpu().numpy()
    fi = FlatIndex(
        dim=d, metric=m, code_kind=_ffi.CODE_BINARY if code_kind == "binary" else _ffi.CODE_RABITQ,
        centroids=centroids, child_offsets=child_offsets, children=np.asarray(children, dtype=np.uint32),
        roots=np.asarray(roots, dtype=np.uint32), vec_offsets=vec_offsets,
        ids=ids_t.cpu().numpy().astype(np.uint64), vectors=vec_lc,
        meta={"branching_factor": branching_factor, "target_leaf_size": target_leaf_size, "seed": seed},
    )
    if code_kind == "binary":
        # threshold is defined over the vectors in the caller's order (binary.rs:127-133)
        # threshold is defined over the vectors in the caller's order (binary.rs:127-133). None: the
        # reference's sequential single-precision fold (it stops growing once the running sum is
        # ~2^24 times a typical element, e.g. beyond ~10^7 values); "mean": the exact mean — what
        # BinaryQuantizer::new(dim, threshold) (binary.rs:114-119) would be given by a careful caller.
        if threshold is None:
            fi.threshold = binary_threshold(x.cpu().numpy())
        elif isinstance(threshold, str):
            assert threshold == "mean"
            fi.threshold = float(x.double().mean().item())
        else:
            fi.threshold = float(threshold)
        if dev.type == "cuda":
            fi.codes = binary_encode_device(x[ids_t], fi.threshold).cpu().numpy().view(np.uint64)
        else:
            fi.codes = pack_binary_codes(vec_lc, fi.threshold)
    elif code_kind == "rabitq":
        rot = make_rotation(d, seed + 7919)
        fi.rabitq_rot_t = rot
        if dev.type == "cuda":
            c, nr, ip = rabitq_encode_device(x[ids_t], torch.from_numpy(rot).to(dev))
            fi.rabitq_codes, fi.rabitq_norm, fi.rabitq_ip_oo = c.cpu().numpy().view(np.uint32), nr.cpu().numpy(), ip.cpu().numpy()
        else:
            fi.rabitq_codes, fi.rabitq_norm, fi.rabitq_ip_oo = rabitq_encode(x[ids_t], torch.from_numpy(rot))
    else:
        raise ValueError(f"unknown code_kind {code_kind!r}")
    return fi.normalised()
