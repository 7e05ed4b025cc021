"""Flat, array-only form of a hierarchical k-means index: what `cuddle_index_create` uploads.

The reference keeps a pointer-rich tree (`ClusterNode`, /root/reference/src/index/hierarchical.rs:18-32,
fields of `ClusteredIndex` :51-70). Here the same information is seven flat arrays plus the
per-vector arrays in *leaf-contiguous order* (leaves in node-id order, each leaf's
`vector_indices` in stored order); see `struct cuddle_index_desc` in include/cuddle_hip.h.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _ffi


def pack_binary_codes(vectors: np.ndarray, threshold: float) -> np.ndarray:
    """BinaryQuantizer::quantize (binary.rs:144-159) for a matrix: bit i set iff v[i] > threshold,
    word i//64, bit i%64. Returns uint64 [n, ceil(d/64)]."""
    n, d = vectors.shape
    words = (d + 63) // 64
    bits = vectors > np.float32(threshold)
    padded = np.zeros((n, words * 64), dtype=np.uint8)
    padded[:, :d] = bits
    packed = np.packbits(padded.reshape(n, words, 8, 8), axis=-1, bitorder="little")  # [n, words, 8, 1]
    return np.ascontiguousarray(packed.reshape(n, words, 8)).view("<u8").reshape(n, words)


@dataclass
class FlatIndex:
    dim: int
    metric: int
    code_kind: int
    centroids: np.ndarray        # f32 [n_nodes, dim]
    child_offsets: np.ndarray    # u32 [n_nodes + 1]
    children: np.ndarray         # u32 [n_children]
    roots: np.ndarray            # u32 [n_roots]
    vec_offsets: np.ndarray      # u64 [n_nodes + 1]
    ids: np.ndarray              # u64 [n_vectors]
    vectors: np.ndarray          # f32 [n_vectors, dim] leaf-contiguous
    threshold: float = 0.0
    codes: Optional[np.ndarray] = None          # u64 [n_vectors, ceil(dim/64)]
    rabitq_rot_t: Optional[np.ndarray] = None   # f32 [dim, dim]
    rabitq_codes: Optional[np.ndarray] = None   # u32 [n_vectors, ceil(dim/32)]
    rabitq_norm: Optional[np.ndarray] = None    # f32 [n_vectors]
    rabitq_ip_oo: Optional[np.ndarray] = None   # f32 [n_vectors]
    meta: dict = field(default_factory=dict)

    # ---- shape helpers ------------------------------------------------------------------
    @property
    def n_nodes(self) -> int:
        return int(self.centroids.shape[0])

    @property
    def n_vectors(self) -> int:
        return int(self.ids.shape[0])

    @property
    def n_roots(self) -> int:
        return int(self.roots.shape[0])

    def is_leaf(self) -> np.ndarray:
        return self.child_offsets[1:] == self.child_offsets[:-1]

    def leaf_sizes(self) -> np.ndarray:
        sizes = (self.vec_offsets[1:] - self.vec_offsets[:-1]).astype(np.int64)
        return sizes[self.is_leaf()]

    def n_levels(self) -> int:
        cur, levels = list(self.roots), 0
        while cur:
            levels += 1
            nxt = []
            for v in cur:
                nxt.extend(self.children[self.child_offsets[v]:self.child_offsets[v + 1]])
            cur = nxt
        return levels

    # ---- normalisation ------------------------------------------------------------------
    def normalised(self) -> "FlatIndex":
        """Same index with every array C-contiguous and of the ABI's dtype."""
        def arr(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)

        return FlatIndex(
            dim=int(self.dim), metric=int(self.metric), code_kind=int(self.code_kind),
            centroids=arr(self.centroids, np.float32).reshape(-1, self.dim),
            child_offsets=arr(self.child_offsets, np.uint32), children=arr(self.children, np.uint32),
            roots=arr(self.roots, np.uint32), vec_offsets=arr(self.vec_offsets, np.uint64),
            ids=arr(self.ids, np.uint64), vectors=arr(self.vectors, np.float32).reshape(-1, self.dim),
            threshold=float(self.threshold), codes=arr(self.codes, np.uint64),
            rabitq_rot_t=arr(self.rabitq_rot_t, np.float32), rabitq_codes=arr(self.rabitq_codes, np.uint32),
            rabitq_norm=arr(self.rabitq_norm, np.float32), rabitq_ip_oo=arr(self.rabitq_ip_oo, np.float32),
            meta=dict(self.meta),
        )

    def desc(self) -> _ffi.IndexDesc:
        """ctypes view for cuddle_index_create; the arrays of `self` must stay alive and must
        already be normalised()."""
        d = _ffi.IndexDesc()
        d.abi_version = _ffi.ABI_VERSION
        d.dim, d.metric, d.code_kind = self.dim, self.metric, self.code_kind
        d.n_nodes, d.n_roots, d.n_vectors = self.n_nodes, self.n_roots, self.n_vectors
        for name in ("centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes",
                     "rabitq_rot_t", "rabitq_codes", "rabitq_norm", "rabitq_ip_oo"):
            a = getattr(self, name)
            if a is not None:
                assert a.flags["C_CONTIGUOUS"], name
            setattr(d, name, None if a is None else C.c_void_p(a.ctypes.data))
        d.threshold = self.threshold
        return d

    # ---- leaf sharding (one shard per GPU) ------------------------------------------------
    def shard(self, rank: int, world: int) -> "FlatIndex":
        """Leaf shard `rank` of `world`: the whole tree (every rank routes identically) with the
        vectors of leaf number j (j-th leaf in node-id order) kept only when j % world == rank.
        Other leaves stay in the tree with an empty vector range."""
        assert 0 <= rank < world
        if world == 1:
            return self
        leaf = self.is_leaf()
        leaf_no = np.cumsum(leaf) - 1
        keep_node = leaf & (leaf_no % world == rank)
        sizes = (self.vec_offsets[1:] - self.vec_offsets[:-1]).astype(np.int64)
        new_sizes = np.where(keep_node, sizes, 0)
        new_off = np.zeros(self.n_nodes + 1, dtype=np.uint64)
        new_off[1:] = np.cumsum(new_sizes)
        keep_vec = np.repeat(keep_node, sizes)

        def sel(a):
            return None if a is None else np.ascontiguousarray(a[keep_vec])

        return FlatIndex(
            dim=self.dim, metric=self.metric, code_kind=self.code_kind, centroids=self.centroids,
            child_offsets=self.child_offsets, children=self.children, roots=self.roots, vec_offsets=new_off,
            ids=sel(self.ids), vectors=sel(self.vectors), threshold=self.threshold, codes=sel(self.codes),
            rabitq_rot_t=self.rabitq_rot_t, rabitq_codes=sel(self.rabitq_codes), rabitq_norm=sel(self.rabitq_norm),
            rabitq_ip_oo=sel(self.rabitq_ip_oo), meta=dict(self.meta, shard=(rank, world)),
        )

    # ---- persistence (our own format; the reference has none, hierarchical.rs:466-481) ----
    def save(self, path: str) -> None:
        arrays = {k: getattr(self, k) for k in (
            "centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes",
            "rabitq_rot_t", "rabitq_codes", "rabitq_norm", "rabitq_ip_oo") if getattr(self, k) is not None}
        np.savez(path, dim=self.dim, metric=self.metric, code_kind=self.code_kind, threshold=np.float32(self.threshold),
                 **arrays)

    _BIN_ARRAYS = (("centroids", np.float32), ("child_offsets", np.uint32), ("children", np.uint32), ("roots", np.uint32),
                   ("vec_offsets", np.uint64), ("ids", np.uint64), ("vectors", np.float32))
    _BIN_RABITQ = (("rabitq_rot_t", np.float32), ("rabitq_codes", np.uint32), ("rabitq_norm", np.float32),
                   ("rabitq_ip_oo", np.float32))

    def save_bin(self, path: str) -> None:
        """The flat-index file the C++ host side reads (include/cuddle_index.hpp, FlatIndex::load):
        "CUDDLEFI", u32 version 1, dim, metric, code_kind, n_nodes, n_roots, n_children, flags (bit 0
        binary codes, bit 1 RaBitQ arrays), f32 threshold, 4 pad bytes, u64 n_vectors, then the arrays
        in struct order, each padded to 8 bytes. Little-endian."""
        x = self.normalised()
        flags = (1 if x.codes is not None else 0) | (2 if x.rabitq_rot_t is not None else 0)
        with open(path, "wb") as f:
            f.write(b"CUDDLEFI")
            f.write(np.array([1, x.dim, x.metric, x.code_kind, x.n_nodes, x.n_roots, x.children.shape[0], flags],
                             dtype="<u4").tobytes())
            f.write(np.array([x.threshold], dtype="<f4").tobytes() + b"\0" * 4)
            f.write(np.array([x.n_vectors], dtype="<u8").tobytes())
            todo = list(self._BIN_ARRAYS) + ([("codes", np.uint64)] if flags & 1 else []) + \
                (list(self._BIN_RABITQ) if flags & 2 else [])
            for name, dt in todo:
                b = np.ascontiguousarray(getattr(x, name), dtype=np.dtype(dt).newbyteorder("<")).tobytes()
                f.write(b + b"\0" * ((8 - len(b) % 8) % 8))

    @staticmethod
    def load_bin(path: str) -> "FlatIndex":
        with open(path, "rb") as f:
            buf = f.read()
        assert buf[:8] == b"CUDDLEFI", "not a flat-index file"
        h = np.frombuffer(buf, dtype="<u4", count=8, offset=8)
        assert h[0] == 1, "unsupported flat-index version"
        dim, metric, code_kind, n_nodes, n_roots, n_children, flags = (int(v) for v in h[1:])
        thr = float(np.frombuffer(buf, dtype="<f4", count=1, offset=40)[0])
        nvec = int(np.frombuffer(buf, dtype="<u8", count=1, offset=48)[0])
        off = 56
        counts = {"centroids": n_nodes * dim, "child_offsets": n_nodes + 1, "children": n_children, "roots": n_roots,
                  "vec_offsets": n_nodes + 1, "ids": nvec, "vectors": nvec * dim, "codes": nvec * ((dim + 63) // 64),
                  "rabitq_rot_t": dim * dim, "rabitq_codes": nvec * ((dim + 31) // 32), "rabitq_norm": nvec,
                  "rabitq_ip_oo": nvec}
        out = {}
        todo = list(FlatIndex._BIN_ARRAYS) + ([("codes", np.uint64)] if flags & 1 else []) + \
            (list(FlatIndex._BIN_RABITQ) if flags & 2 else [])
        for name, dt in todo:
            n = counts[name]
            out[name] = np.frombuffer(buf, dtype=np.dtype(dt).newbyteorder("<"), count=n, offset=off).copy()
            nbytes = n * np.dtype(dt).itemsize
            off += nbytes + (8 - nbytes % 8) % 8
        words, w32 = (dim + 63) // 64, (dim + 31) // 32
        shaped = {"centroids": (-1, dim), "vectors": (-1, dim), "codes": (-1, words), "rabitq_rot_t": (dim, dim),
                  "rabitq_codes": (-1, w32)}
        for k2, shp in shaped.items():
            if k2 in out:
                out[k2] = out[k2].reshape(shp)
        return FlatIndex(dim=dim, metric=metric, code_kind=code_kind, threshold=thr, **out).normalised()

    @staticmethod
    def load(path: str) -> "FlatIndex":
        z = np.load(path, allow_pickle=False)
        g = lambda k: z[k] if k in z.files else None  # noqa: E731
        return FlatIndex(
            dim=int(z["dim"]), metric=int(z["metric"]), code_kind=int(z["code_kind"]), centroids=z["centroids"],
            child_offsets=z["child_offsets"], children=z["children"], roots=z["roots"], vec_offsets=z["vec_offsets"],
            ids=z["ids"], vectors=z["vectors"], threshold=float(z["threshold"]), codes=g("codes"),
            rabitq_rot_t=g("rabitq_rot_t"), rabitq_codes=g("rabitq_codes"), rabitq_norm=g("rabitq_norm"),
            rabitq_ip_oo=g("rabitq_ip_oo"),
        ).normalised()
