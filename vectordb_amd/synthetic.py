"""Index-by-construction for the large synthetic shards (100M - 1B vector configs): a hierarchical
Gaussian whose tree is known from the generator, produced directly on the GPU in leaf-contiguous
order and handed to the library as device arrays (CUDDLE_DESC_DEVICE_ARRAYS | ADOPT_VECTORS), so a
64 GB shard is never copied through the host. k-means is not run: the reference's build takes
837 s for 1M x 1024-d (/root/reference/benches/saved/jan_29_1M_vectors/shell_output.txt:27) and is
not part of the query path; any tree with centroids is a valid input of `search`.

Tree: `levels` levels of `fanout`-way nodes (level 0 = roots), the last level's nodes are leaves of
`leaf_size` vectors each. Node ids are breadth-first. Leaf centroid = mean of its vectors, inner
centroid = mean of its children's centroids.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _ffi
from .flat_index import FlatIndex


@dataclass
class DeviceTreeIndex:
    dim: int
    metric: int
    threshold: float
    child_offsets: np.ndarray   # host u32
    children: np.ndarray        # host u32
    roots: np.ndarray           # host u32
    vec_offsets: np.ndarray     # host u64
    centroids: torch.Tensor     # device f32 [n_nodes, dim]
    ids: torch.Tensor           # device i64 [N]
    vectors: torch.Tensor       # device f32 [N, dim]
    codes: torch.Tensor = None  # device i64 [N, W]           (binary codes)
    code_kind: int = _ffi.CODE_BINARY
    rot_t: torch.Tensor = None      # device f32 [dim, dim]   (RaBitQ: rotation_matrix_t)
    rq_codes: torch.Tensor = None   # device i32 [N, ceil(dim/32)] (bit pattern of u32 words)
    rq_norm: torch.Tensor = None    # device f32 [N]
    rq_ipoo: torch.Tensor = None    # device f32 [N]

    @property
    def n_vectors(self) -> int:
        return int(self.ids.shape[0])

    def desc(self) -> _ffi.IndexDesc:
        d = _ffi.IndexDesc()
        d.abi_version = _ffi.ABI_VERSION
        d.dim, d.metric, d.code_kind = self.dim, self.metric, self.code_kind
        d.n_nodes, d.n_roots, d.n_vectors = int(self.centroids.shape[0]), int(self.roots.shape[0]), self.n_vectors
        d.child_offsets = C.c_void_p(self.child_offsets.ctypes.data)
        d.children = C.c_void_p(self.children.ctypes.data)
        d.roots = C.c_void_p(self.roots.ctypes.data)
        d.vec_offsets = C.c_void_p(self.vec_offsets.ctypes.data)
        d.centroids = C.c_void_p(self.centroids.data_ptr())
        d.ids = C.c_void_p(self.ids.data_ptr())
        d.vectors = C.c_void_p(self.vectors.data_ptr())
        if self.code_kind == _ffi.CODE_BINARY:
            d.codes = C.c_void_p(self.codes.data_ptr())
        else:
            d.rabitq_rot_t = C.c_void_p(self.rot_t.data_ptr())
            d.rabitq_codes = C.c_void_p(self.rq_codes.data_ptr())
            d.rabitq_norm = C.c_void_p(self.rq_norm.data_ptr())
            d.rabitq_ip_oo = C.c_void_p(self.rq_ipoo.data_ptr())
        d.threshold = self.threshold
        d.flags = _ffi.DESC_DEVICE_ARRAYS | _ffi.DESC_ADOPT_VECTORS
        return d

    def create(self, device_id: int):
        from .index import DeviceIndex

        torch.cuda.synchronize()
        return DeviceIndex(None, device_id, desc=self.desc(), keepalive=self)

    def to_flat(self) -> FlatIndex:
        """Host copy (small cases: lets the oracle check the same index)."""
        rq = {}
        if self.code_kind == _ffi.CODE_RABITQ:
            rq = dict(rabitq_rot_t=self.rot_t.cpu().numpy(), rabitq_codes=self.rq_codes.cpu().numpy().view(np.uint32),
                      rabitq_norm=self.rq_norm.cpu().numpy(), rabitq_ip_oo=self.rq_ipoo.cpu().numpy())
        return FlatIndex(dim=self.dim, metric=self.metric, code_kind=self.code_kind,
                         centroids=self.centroids.cpu().numpy(), child_offsets=self.child_offsets, children=self.children,
                         roots=self.roots, vec_offsets=self.vec_offsets, ids=self.ids.cpu().numpy().astype(np.uint64),
                         vectors=self.vectors.cpu().numpy(), threshold=self.threshold,
                         codes=None if self.codes is None else self.codes.cpu().numpy().view(np.uint64), **rq).normalised()


def _pack_codes(x: torch.Tensor, threshold: float) -> torch.Tensor:
    """BinaryQuantizer::quantize of the rows: through the library for CUDA rows
    (cuddle_binary_encode_device); CPU rows (the generator's gloo rehearsals) in torch."""
    if x.is_cuda:
        from .builder import binary_encode_device

        return binary_encode_device(x, threshold)
    from .builder import pack_binary_codes_torch

    return torch.from_numpy(pack_binary_codes_torch(x, threshold).view(np.int64))


def _rabitq_encode_device(vectors: torch.Tensor, rot_t: torch.Tensor):
    """RaBitQQuantizer::quantize_batch of the rows: through the library for CUDA rows
    (cuddle_rabitq_encode_device); CPU rows in torch (builder.rabitq_encode)."""
    if vectors.is_cuda:
        from .builder import rabitq_encode_device

        return rabitq_encode_device(vectors, rot_t)
    from .builder import rabitq_encode

    c, n, i = rabitq_encode(vectors, rot_t)
    return torch.from_numpy(c.view(np.int32)), torch.from_numpy(n), torch.from_numpy(i)


def build_tree_index_on_device(dim: int, fanout: int, levels: int, leaf_size: int, device, seed: int = 0,
                               sigmas=(0.35, 0.22, 0.12), metric: int = _ffi.METRIC_L2, style: str = "sift",
                               n_queries: int = 0, chunk_leaves: int = 8192, code_kind: int = _ffi.CODE_BINARY,
                               rank: int = 0, world: int = 1, group=None):
    """Returns (DeviceTreeIndex, queries [n_queries, dim]).
    sigmas: spread of each level's centres around its parent, last entry = spread of the points
    around their leaf centre. style "sift": values mapped to non-negative integers in the u8 range.

    world == 1: the whole index, N = fanout**levels * leaf_size vectors.
    world > 1 (one process per GPU, torch.distributed initialised): the index has `world` times as
    many first-level nodes (N = world * fanout**levels * leaf_size) and this call returns leaf shard
    `rank` of it: the whole tree with every centroid (every rank must route identically), and the
    vectors, codes and ids of the leaves j with j % world == rank only, generated here and nowhere
    else. The ranks exchange what the tree needs - each leaf's centroid (an all-gather) and the sum
    of all values for the binary threshold (an all-reduce); queries are the same on every rank."""
    assert levels >= 1 and len(sigmas) >= 2 and 0 <= rank < world
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    widths = [world * fanout ** (l + 1) for l in range(levels)]
    offs = np.concatenate([[0], np.cumsum(widths)]).astype(np.int64)
    n_nodes, n_leaves = int(offs[-1]), widths[-1]
    # generating centres, level by level (the same on every rank)
    centres = [torch.rand((widths[0], dim), generator=g, device=device) * 2 - 1]
    for l in range(1, levels):
        s = sigmas[min(l - 1, len(sigmas) - 2)]
        parent = centres[-1].repeat_interleave(fanout, dim=0)
        centres.append(parent + s * torch.randn((widths[l], dim), generator=g, device=device))
    sig_pts = sigmas[-1]
    # the points of a rank come from a stream of their own, so that the shared stream stays in step
    gp = g
    if world > 1:
        gp = torch.Generator(device=device)
        gp.manual_seed(seed + 7919 * (rank + 1))

    def finish(x):
        if style == "sift":
            return torch.clamp(torch.round((x + 1.8) * 50.0), 0, 255)
        return x

    own = torch.arange(rank, n_leaves, world, device=device)  # this rank's leaves, ascending node order
    n_own = int(own.shape[0])
    n_vec = n_own * leaf_size
    vectors = torch.empty((n_vec, dim), dtype=torch.float32, device=device)
    own_cent = torch.empty((n_own, dim), dtype=torch.float32, device=device)
    total = torch.zeros((), dtype=torch.float64, device=device)
    for a in range(0, n_own, chunk_leaves):
        b = min(a + chunk_leaves, n_own)
        x = centres[-1][own[a:b], None, :] + sig_pts * torch.randn((b - a, leaf_size, dim), generator=gp, device=device)
        x = finish(x)
        own_cent[a:b] = x.mean(dim=1)
        total += x.double().sum()
        vectors[a * leaf_size:b * leaf_size] = x.reshape(-1, dim)
    if world > 1:
        import torch.distributed as dist
        # RCCL works on the device buffers; a gloo rehearsal (several ranks on one GPU) goes through the host
        where = "cpu" if dist.get_backend(group) == "gloo" else device
        gathered = torch.empty((world, n_own, dim), dtype=torch.float32, device=where)
        dist.all_gather_into_tensor(gathered.view(world * n_own, dim), own_cent.to(where), group=group)
        leaf_cent = gathered.to(device).permute(1, 0, 2).reshape(n_leaves, dim).contiguous()  # leaf j = i * world + r
        del gathered
        total_h = total.to(where)
        dist.all_reduce(total_h, group=group)
        total = total_h.to(device)
    else:
        leaf_cent = own_cent
    threshold = float((total / (n_leaves * leaf_size * dim)).item())
    codes, rq = None, {}
    if code_kind == _ffi.CODE_BINARY:
        codes = _pack_codes(vectors, threshold)
    else:
        from .builder import make_rotation
        rot_t = torch.from_numpy(make_rotation(dim, seed + 17)).to(device)
        rc, rn, ri = _rabitq_encode_device(vectors, rot_t)
        rq = dict(rot_t=rot_t, rq_codes=rc, rq_norm=rn, rq_ipoo=ri)
    # centroids bottom-up
    cents = [None] * levels
    cents[-1] = leaf_cent
    for l in range(levels - 2, -1, -1):
        cents[l] = cents[l + 1].view(widths[l], fanout, dim).mean(dim=1)
    centroids = torch.cat(cents, dim=0).contiguous()
    del cents, leaf_cent
    # tree arrays (host)
    child_offsets = np.zeros(n_nodes + 1, dtype=np.uint32)
    counts = np.zeros(n_nodes, dtype=np.int64)
    counts[:offs[levels - 1]] = fanout
    child_offsets[1:] = np.cumsum(counts)
    children = np.arange(offs[1], n_nodes, dtype=np.uint32) if levels > 1 else np.zeros(0, dtype=np.uint32)
    vec_counts = np.zeros(n_nodes, dtype=np.int64)
    vec_counts[offs[levels - 1] + rank::world] = leaf_size
    vec_offsets = np.zeros(n_nodes + 1, dtype=np.uint64)
    vec_offsets[1:] = np.cumsum(vec_counts)
    roots = np.arange(widths[0], dtype=np.uint32)
    # id of a vector = its place in the whole index: leaf * leaf_size + place in the leaf
    ids = (own[:, None] * leaf_size + torch.arange(leaf_size, device=device)[None, :]).reshape(-1).to(torch.int64)
    queries = torch.empty((0, dim), device=device)
    if n_queries:
        leaf = torch.randint(0, n_leaves, (n_queries,), generator=g, device=device)
        queries = finish(centres[-1][leaf] + sig_pts * torch.randn((n_queries, dim), generator=g, device=device))
    idx = DeviceTreeIndex(dim=dim, metric=metric, threshold=threshold, child_offsets=child_offsets, children=children,
                          roots=roots, vec_offsets=vec_offsets, centroids=centroids, ids=ids, vectors=vectors, codes=codes,
                          code_kind=code_kind, **rq)
    return idx, queries.contiguous()
