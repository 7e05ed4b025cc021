"""A VectorDBBench client over `PyVectorDB`: the class a benchmark harness of that kind drives
(docs/vectordbbench_integration/VECTORDBBENCH_INTEGRATION.md:59-108 and :356-390 describe the reference's:
the harness' `VectorDB` interface on top of the Python module, an ID mapping between the harness' integer ids
and array indices, and pickling that carries the configuration only, the index being rebuilt from the staging
file in the process that unpickles).

The interface is the harness' own: `__init__(dim, db_config, db_case_config, collection_name, drop_old)`,
`init()` as a context manager, `insert_embeddings(embeddings, metadata) -> (count, error)`, `optimize()`,
`search_embedding(query, k) -> [ids]`. The harness package itself is not needed: when it is importable the class
registers as a subclass of its `VectorDB`, otherwise it stands alone. `search_embeddings(queries, k)` is the
batched form the GPU path is made for.

Files: `<path>.staging` (PyVectorDB's, python_bindings.rs:313-430) and `<path>.idmap.npy`, the harness ids in
insertion order as a plain int64 array (the reference pickles a dict; an array needs no unpickling).
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import asdict, dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .pyvectordb import PyVectorDB

try:  # the harness, where it is installed
    from vectordb_bench.backend.clients.api import VectorDB as _HarnessVectorDB  # type: ignore
except Exception:  # noqa: BLE001 - any failure to import it means "not installed here"
    _HarnessVectorDB = object


@dataclass
class CuddleCaseConfig:
    """Build and search parameters, named as the reference's command line names them
    (VECTORDBBENCH_INTEGRATION.md:130-138): --branching-factor, --target-leaf-size, --probes, --rerank-factor."""
    branching_factor: int = 100
    target_leaf_size: int = 100
    probes: int = 100
    rerank_factor: int = 10
    metric: str = "Cosine"  # 'L2', 'Cosine' or 'InnerProduct' (PyVectorDB's names)

    def index_param(self) -> Dict[str, int]:
        return {"branching_factor": self.branching_factor, "target_leaf_size": self.target_leaf_size}

    def search_param(self) -> Dict[str, int]:
        return {"probes": self.probes, "rerank_factor": self.rerank_factor}


class CuddleVectorDB(_HarnessVectorDB):
    def __init__(self, dim: int, db_config: Optional[dict] = None, db_case_config: Optional[CuddleCaseConfig] = None,
                 collection_name: str = "vectordb_bench", drop_old: bool = False, **kwargs):
        cfg = dict(db_config or {})
        self.dim = int(dim)
        self.case_config = db_case_config or CuddleCaseConfig()
        self.path = cfg.get("path", f"/tmp/{collection_name}.bin")
        self.device_id = int(cfg.get("device_id", 0))
        self.seed = int(cfg.get("seed", 0))
        self.db: Optional[PyVectorDB] = None
        self._ids: List[np.ndarray] = []  # harness ids of the rows inserted by this object, in order
        self._idmap: Optional[np.ndarray] = None  # array index -> harness id, once optimize() / init() has it
        if drop_old:
            for p in (self._staging_path, self._idmap_path):
                with contextlib.suppress(FileNotFoundError):
                    os.remove(p)

    # -- files -----------------------------------------------------------------------------------------------
    @property
    def _staging_path(self) -> str:
        return f"{self.path}.staging"

    @property
    def _idmap_path(self) -> str:
        return f"{self.path}.idmap.npy"

    def _new_db(self) -> PyVectorDB:
        c = self.case_config
        return PyVectorDB(self.dim, c.branching_factor, c.target_leaf_size, c.metric, self.path, self.device_id, self.seed)

    # -- the harness' interface ------------------------------------------------------------------------------
    @contextlib.contextmanager
    def init(self):
        """Entered once per process that loads or searches. A process that did not see the inserts (it got this
        object by unpickling) finds the staging file and the id mapping and rebuilds the index from them."""
        if self.db is None:
            self.db = self._new_db()
            if os.path.exists(self._idmap_path) and os.path.exists(self._staging_path) and not self._ids:
                self._idmap = np.load(self._idmap_path, allow_pickle=False)
                self.db.optimize()  # no rows held: reads the staging file
                if len(self.db) != self._idmap.shape[0]:
                    raise RuntimeError(f"id mapping has {self._idmap.shape[0]} entries, the staging file {len(self.db)} vectors")
        yield

    def need_normalize_cosine(self) -> bool:
        return False  # the cosine metric normalises by itself (src/distance/mod.rs:23-34)

    def insert_embeddings(self, embeddings: Iterable[Sequence[float]], metadata: Sequence[int], **kwargs) -> Tuple[int, Optional[Exception]]:
        """Rows and their harness ids; (rows inserted, error or None) as the harness expects."""
        if self.db is None:
            self.db = self._new_db()
        try:
            rows = np.ascontiguousarray(embeddings, dtype=np.float32)
            ids = np.asarray(metadata, dtype=np.int64).reshape(-1)
            if rows.ndim != 2 or rows.shape[0] != ids.shape[0]:
                raise ValueError(f"{rows.shape[0] if rows.ndim == 2 else 0} embeddings for {ids.shape[0]} ids")
            n = self.db.insert_embeddings(rows)
            self._ids.append(ids)
            return n, None
        except Exception as e:  # noqa: BLE001 - the harness wants the error returned, not raised
            return 0, e

    def optimize(self, data_size: Optional[int] = None) -> None:
        """Builds the index over everything inserted and saves the id mapping beside the staging file."""
        if self.db is None:
            raise RuntimeError("optimize() before init() / insert_embeddings()")
        if self._ids:
            self._idmap = np.concatenate(self._ids)
            self._ids = []
            np.save(self._idmap_path, self._idmap, allow_pickle=False)
        self.db.optimize()
        if self._idmap is None or self._idmap.shape[0] != len(self.db):
            raise RuntimeError("No ID mapping found for the indexed vectors")

    def _map(self, idx: np.ndarray) -> np.ndarray:
        if self._idmap is None:
            raise RuntimeError("Index not optimized")
        return self._idmap[idx]

    def search_embedding(self, query: Sequence[float], k: int = 100, filters: Optional[dict] = None, **kwargs) -> List[int]:
        """Harness ids of the k nearest, nearest first."""
        if self.db is None or self.db.index is None:
            raise RuntimeError("Index not optimized")
        c = self.case_config
        res = self.db.search(query, k, c.probes, c.rerank_factor)
        return [int(i) for i in self._map(np.fromiter((r[0] for r in res), dtype=np.int64, count=len(res)))]

    def search_embeddings(self, queries, k: int = 100) -> List[List[int]]:
        """search_embedding for every row of `queries` in one GPU batch."""
        if self.db is None or self.db.index is None:
            raise RuntimeError("Index not optimized")
        c = self.case_config
        q = np.ascontiguousarray(queries, dtype=np.float32)
        ids, _dist, count = self.db.index.search_batch(q, k, c.probes, c.rerank_factor)
        return [[int(v) for v in self._map(ids[i, :int(count[i])].astype(np.int64))] for i in range(q.shape[0])]

    # -- pickling: the configuration travels, the index does not (VECTORDBBENCH_INTEGRATION.md:358-373) ---------
    def __getstate__(self) -> dict:
        return {"dim": self.dim, "case_config": asdict(self.case_config), "path": self.path, "device_id": self.device_id,
                "seed": self.seed}

    def __setstate__(self, st: dict) -> None:
        self.dim = st["dim"]
        self.case_config = CuddleCaseConfig(**st["case_config"])
        self.path, self.device_id, self.seed = st["path"], st["device_id"], st["seed"]
        self.db, self._ids, self._idmap = None, [], None
