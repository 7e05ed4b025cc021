"""`PyVectorDB`, the class the reference's Python module exports (src/python_bindings.rs:27-307),
with the same constructor, methods, defaults, return shapes and error messages, running its
searches on the GPU. Vectors are batched with insert_embeddings(), optimize() builds a
ClusteredIndexWithRaBitQ (20 k-means iterations, :146-153) and uploads it; search() and
search_with_stats() are single-query calls as in the reference, search_batch() is the batched form
the GPU path is made for. The staging file `<temp_file_path>.staging` follows :313-430, so that a
process that did not see the inserts can still optimize()."""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .datasets import append_staging, read_staging
from .index import ClusteredIndexWithRaBitQ

__version__ = "0.1.0"

_METRICS = {"L2": "l2", "Cosine": "cosine", "InnerProduct": "dot", "DotProduct": "dot"}


class PyVectorDB:
    def __init__(self, dimension: int, branching_factor: int = 100, target_leaf_size: int = 100, metric: str = "L2",
                 temp_file_path: str = "/tmp/vectordb_py.bin", device_id: int = 0, seed: int = 0):
        if metric not in _METRICS:
            raise ValueError(f"Unknown metric '{metric}'. Use 'L2', 'Cosine', or 'InnerProduct'")
        self.dimension = int(dimension)
        self.branching_factor = int(branching_factor)
        self.target_leaf_size = int(target_leaf_size)
        self.metric = _METRICS[metric]
        self.temp_file_path = temp_file_path
        self.device_id = device_id
        self.seed = seed
        self.index: Optional[ClusteredIndexWithRaBitQ] = None
        self._pending: List[np.ndarray] = []
        self._pending_rows = 0

    @property
    def _staging_path(self) -> str:
        return f"{self.temp_file_path}.staging"

    def insert_embeddings(self, vectors: Sequence[Sequence[float]]) -> int:
        """:88-113 — validates every row's length, keeps the rows and appends them to the staging file."""
        rows = []
        for i, v in enumerate(vectors):
            r = np.asarray(v, dtype=np.float32).reshape(-1)
            if r.shape[0] != self.dimension:
                raise ValueError(f"Vector {i} has dimension {r.shape[0]}, expected {self.dimension}")
            rows.append(r)
        if rows:
            block = np.stack(rows)
            self._pending.append(block)
            self._pending_rows += block.shape[0]
        if self._pending_rows:  # :321-323: nothing held -> the file is not touched
            # :326-369: the file's own count says how many of the held rows it already has
            have = 0
            if os.path.exists(self._staging_path):
                have = int(np.fromfile(self._staging_path, dtype="<u8", count=1)[0])
            held = np.concatenate(self._pending) if len(self._pending) > 1 else self._pending[0]
            self._pending = [held]
            append_staging(self._staging_path, held[have:], self.dimension)
            if have > held.shape[0]:  # :372-376 writes the held count whatever the file said
                with open(self._staging_path, "r+b") as f:
                    np.asarray([held.shape[0]], dtype="<u8").tofile(f)
        return len(rows)

    def optimize(self) -> float:
        """:122-166 — builds from the held rows, or from the staging file when none are held; returns seconds."""
        if self._pending_rows:
            vectors = np.concatenate(self._pending)
            self._pending, self._pending_rows = [], 0
        else:
            vectors = read_staging(self._staging_path, self.dimension)
            if vectors.shape[0] == 0:
                raise ValueError("No vectors to build index from")
            print(f"Loaded {vectors.shape[0]} vectors from staging file")
        print(f"Building index with {vectors.shape[0]} vectors...")
        t0 = time.perf_counter()
        try:
            self.index = ClusteredIndexWithRaBitQ.build(vectors, self.temp_file_path, self.branching_factor,
                                                        self.target_leaf_size, self.metric, 20, seed=self.seed,
                                                        device_id=self.device_id)
        except OSError as e:
            raise RuntimeError(f"Failed to build index: {e}") from e
        dt = time.perf_counter() - t0
        print(f"Index built in {dt:.2f}s")
        return dt

    def _built(self, msg: str = "Index not built. Call optimize() first") -> ClusteredIndexWithRaBitQ:
        if self.index is None:
            raise RuntimeError(msg)
        return self.index

    def _check_query(self, n: int) -> None:
        if n != self.dimension:
            raise ValueError(f"Query dimension {n} doesn't match index dimension {self.dimension}")

    def search(self, query: Sequence[float], k: int = 10, probes: int = 100, rerank_factor: int = 10) -> List[Tuple[int, float]]:
        """:178-204"""
        q = np.asarray(query, dtype=np.float32).reshape(-1)
        self._check_query(q.shape[0])
        return self._built().search(q, k, probes, rerank_factor)

    def search_batch(self, queries, k: int = 10, probes: int = 100, rerank_factor: int = 10):
        """Not in the reference: search() for every row in one GPU batch; a list of result lists."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        q = q.reshape(-1, q.shape[-1]) if q.size else q.reshape(0, self.dimension)
        self._check_query(q.shape[1])
        ids, dist, count = self._built().search_batch(q, k, probes, rerank_factor)
        return [[(int(ids[i, j]), float(dist[i, j])) for j in range(int(count[i]))] for i in range(q.shape[0])]

    def search_with_stats(self, query: Sequence[float], k: int = 10, probes: int = 100, rerank_factor: int = 10):
        """:218-253 — (results, dict) with the six keys the reference puts in the dict."""
        q = np.asarray(query, dtype=np.float32).reshape(-1)
        self._check_query(q.shape[0])
        results, st = self._built().search_with_stats(q, k, probes, rerank_factor)
        return results, {
            "total_leaves": st.total_leaves, "leaves_searched": st.leaves_searched, "total_vectors": st.total_vectors,
            "vectors_scanned_rabitq": st.vectors_scanned_binary, "vectors_reranked_full": st.vectors_reranked_full,
            "tree_depth": st.tree_depth}

    def get_stats(self) -> Dict[str, int]:
        """:259-275"""
        ix = self._built("Index not built")
        fi = ix.flat
        leaf = fi.is_leaf()
        sizes = (fi.vec_offsets[1:] - fi.vec_offsets[:-1])[leaf]
        code_words = fi.rabitq_codes.shape[1]
        return {
            "num_vectors": len(ix), "dimension": fi.dim, "max_depth": max(ix.dev.levels - 1, 0), "num_nodes": fi.n_nodes,
            "num_leaves": int(leaf.sum()), "max_leaf_size": int(sizes.max()) if sizes.size else 0,
            # hierarchical_with_rabitq.rs:479-494: size_of::<ClusterNodeWithRaBitQ>() (a usize and three Vecs, 80
            # bytes on a 64-bit target) per node + per vector code.len() * 8 + 4 + 4; :497-499: the vector file's size
            "memory_usage_bytes": 80 * fi.n_nodes + len(ix) * (code_words * 8 + 8),
            "disk_usage_bytes": len(ix) * fi.dim * 4}

    def __len__(self) -> int:
        """:278-287"""
        return len(self.index) if self.index is not None else self._pending_rows

    def __repr__(self) -> str:
        """:290-306"""
        if self.index is None:
            return f"PyVectorDB(dimension={self.dimension}, not_built, accumulated_vectors={self._pending_rows})"
        st = self.get_stats()
        return (f"PyVectorDB(vectors={st['num_vectors']}, dimension={self.dimension}, depth={st['max_depth']}, "
                f"leaves={st['num_leaves']})")
