"""Data sources of the benchmark configs: readers for the reference's on-disk formats and
seeded synthetic stand-ins of the same shape (no dataset ships with the reference and none can
be downloaded here; synthetic data is always labelled as such).

Formats: .fvecs/.ivecs — per vector a little-endian i32 dimension then that many f32/i32
(/root/reference/datasets/sift/loader.rs:10-87); generator .bin — [u32 n][u32 d][f32 ...]
(/root/reference/datasets/generate_datasets.rs:165-189); raw vector file — count*dim f32, no
header (/root/reference/src/storage/mmap.rs:26-98); staging file of the Python binding —
[u64 n][u64 d][f32 ...], grown by appending (/root/reference/src/python_bindings.rs:313-430).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np


def read_fvecs(path: str, limit: Optional[int] = None) -> np.ndarray:
    raw = np.fromfile(path, dtype="<i4")
    if raw.size == 0:
        return np.zeros((0, 0), dtype=np.float32)
    d = int(raw[0])
    rows = raw.reshape(-1, d + 1)
    if not np.all(rows[:, 0] == d):
        raise ValueError(f"{path}: inconsistent dimension headers")
    if limit is not None:
        rows = rows[:limit]
    return np.ascontiguousarray(rows[:, 1:]).view("<f4")


def read_ivecs(path: str, limit: Optional[int] = None) -> np.ndarray:
    raw = np.fromfile(path, dtype="<i4")
    if raw.size == 0:
        return np.zeros((0, 0), dtype=np.int32)
    d = int(raw[0])
    rows = raw.reshape(-1, d + 1)
    if not np.all(rows[:, 0] == d):
        raise ValueError(f"{path}: inconsistent dimension headers")
    if limit is not None:
        rows = rows[:limit]
    return np.ascontiguousarray(rows[:, 1:])


def write_fvecs(path: str, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, dtype="<f4")
    out = np.empty((x.shape[0], x.shape[1] + 1), dtype="<i4")
    out[:, 0] = x.shape[1]
    out[:, 1:] = x.view("<i4")
    out.tofile(path)


def write_ivecs(path: str, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, dtype="<i4")
    out = np.empty((x.shape[0], x.shape[1] + 1), dtype="<i4")
    out[:, 0] = x.shape[1]
    out[:, 1:] = x
    out.tofile(path)


def read_generator_bin(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        n, d = np.fromfile(f, dtype="<u4", count=2)
        return np.fromfile(f, dtype="<f4", count=int(n) * int(d)).reshape(int(n), int(d))


def write_generator_bin(path: str, x: np.ndarray) -> None:
    with open(path, "wb") as f:
        np.asarray(x.shape, dtype="<u4").tofile(f)
        np.ascontiguousarray(x, dtype="<f4").tofile(f)


def read_raw_vectors(path: str, dim: int, count: int) -> np.ndarray:
    """MmapVectorStore::open (mmap.rs:26-52): size must be exactly count*dim*4 bytes."""
    if os.path.getsize(path) != count * dim * 4:
        raise ValueError(f"File size mismatch: expected {count * dim * 4} bytes, got {os.path.getsize(path)}")
    return np.memmap(path, dtype="<f4", mode="r", shape=(count, dim))


def append_staging(path: str, new_rows: np.ndarray, dim: int) -> int:
    """save_vectors_to_staging (python_bindings.rs:313-382): the first call creates the file with
    the header [u64 count = 0][u64 dim]; every call appends its rows as little-endian f32 and then
    rewrites the count at offset 0. Returns the count now in the header."""
    new_rows = np.ascontiguousarray(new_rows, dtype="<f4").reshape(-1, dim)
    if os.path.exists(path):
        with open(path, "rb") as f:
            before, have_dim = (int(v) for v in np.fromfile(f, dtype="<u8", count=2))
        if have_dim != dim:
            raise ValueError(f"Dimension mismatch: expected {dim}, got {have_dim}")
    else:
        before = 0
        with open(path, "wb") as f:
            np.asarray([0, dim], dtype="<u8").tofile(f)
    with open(path, "r+b") as f:
        f.seek(0, os.SEEK_END)
        new_rows.tofile(f)
        f.seek(0)
        np.asarray([before + new_rows.shape[0]], dtype="<u8").tofile(f)
    return before + new_rows.shape[0]


def read_staging(path: str, dim: int) -> np.ndarray:
    """load_vectors_from_staging (python_bindings.rs:385-430): no file -> no vectors; a header whose
    dimension differs from `dim` is an error; a file shorter than its header promises is an error
    (read_exact fails there)."""
    if not os.path.exists(path):
        return np.zeros((0, dim), dtype=np.float32)
    with open(path, "rb") as f:
        head = np.fromfile(f, dtype="<u8", count=2)
        if head.size != 2:
            raise ValueError("Failed to read header")
        n, d = int(head[0]), int(head[1])
        if d != dim:
            raise ValueError(f"Dimension mismatch: expected {dim}, got {d}")
        body = np.fromfile(f, dtype="<f4", count=n * d)
    if body.size != n * d:
        raise ValueError("Failed to read vector data")
    return body.reshape(n, d)


def synthetic_mixture(n: int, dim: int, seed: int, device="cpu", n_top: int = 100, n_sub: int = 10,
                      sigma_sub: float = 0.30, sigma: float = 0.25, style: str = "gist", n_queries: int = 0,
                      latent_dim: int = 0, ambient_noise: float = 0.02, query_block: int = 0):
    """Two-level Gaussian mixture on `device` (torch): n_top centres U(-1,1), n_sub sub-centres
    around each (sigma_sub), points around sub-centres (sigma). With latent_dim > 0 the mixture
    lives in R^latent_dim and is embedded in R^dim by a fixed random linear map plus a little
    isotropic noise — descriptor data has a low intrinsic dimension, and without one every
    neighbour of a point is (nearly) equally far away. style "sift" maps the result to
    non-negative integer-valued f32 in the u8 range; "gist" to small non-negative reals. Returns
    (base [n,dim], queries [n_queries,dim]) float32 tensors; queries are held-out draws. With
    query_block > 0 the queries are drawn query_block at a time, so a longer query set starts with
    exactly the queries of a shorter one (the generator's output depends on the size of each draw)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m_dim = latent_dim if latent_dim > 0 else dim
    top = torch.rand((n_top, m_dim), generator=g, device=device) * 2 - 1
    sub = top[:, None, :] + sigma_sub * torch.randn((n_top, n_sub, m_dim), generator=g, device=device)
    sub = sub.reshape(n_top * n_sub, m_dim)
    proj = None
    if latent_dim > 0:
        proj = torch.randn((m_dim, dim), generator=g, device=device) / (m_dim ** 0.5)

    def draw(m, chunk=1 << 18):
        out = torch.empty((m, dim), dtype=torch.float32, device=device)
        for s in range(0, m, chunk):
            e = min(s + chunk, m)
            comp = torch.randint(0, sub.shape[0], (e - s,), generator=g, device=device)
            z = sub[comp] + sigma * torch.randn((e - s, m_dim), generator=g, device=device)
            if proj is not None:
                z = z @ proj + ambient_noise * torch.randn((e - s, dim), generator=g, device=device)
            out[s:e] = z
        if style == "sift":
            out = torch.clamp(torch.round((out + 1.6) * 55.0), 0, 255)
        elif style == "gist":
            out = torch.clamp((out + 1.6) * 0.25, min=0)
        return out

    base = draw(n)
    if not n_queries:
        queries = torch.empty((0, dim), device=device)
    elif query_block > 0 and n_queries > query_block:
        queries = torch.cat([draw(min(query_block, n_queries - s)) for s in range(0, n_queries, query_block)])
    else:
        queries = draw(n_queries)
    return base, queries


def brute_force_topk(base, queries, k: int, metric: str = "l2", chunk: int = 1 << 18, return_dist: bool = False):
    """Exact ground truth on the GPU (torch): row numbers [nq,k] of the k nearest base rows
    (and their distances with return_dist, for merging the lists of several shards)."""
    import torch

    nq = queries.shape[0]
    best_d = torch.full((nq, k), float("inf"), device=queries.device)
    best_i = torch.full((nq, k), -1, dtype=torch.long, device=queries.device)
    qn = (queries * queries).sum(1, keepdim=True)
    for s in range(0, base.shape[0], chunk):
        b = base[s:s + chunk]
        dots = queries @ b.t()
        if metric == "l2":
            d = qn + (b * b).sum(1)[None, :] - 2 * dots
        elif metric == "dot":
            d = -dots
        else:
            d = 1 - dots / (qn.sqrt() * (b * b).sum(1).sqrt()[None, :]).clamp_min(1e-30)
        cd = torch.cat([best_d, d], dim=1)
        ci = torch.cat([best_i, torch.arange(s, s + b.shape[0], device=b.device)[None, :].expand(nq, -1)], dim=1)
        best_d, sel = torch.topk(cd, k, dim=1, largest=False)
        best_i = torch.gather(ci, 1, sel)
    return (best_i, best_d) if return_dist else best_i


def recall_at_k(found_ids: np.ndarray, gt_ids: np.ndarray, k: int) -> float:
    """benches/sift_benchmark.rs:26-32: |GT[:k] ∩ result[:k]| / k, averaged over the queries."""
    hits = 0
    for f, g in zip(found_ids, gt_ids):
        hits += len(set(int(v) for v in g[:k]) & set(int(v) for v in f[:k]))
    return hits / (k * len(gt_ids))
