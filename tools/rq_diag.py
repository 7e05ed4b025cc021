"""RaBitQ scan diagnostics on a bench configuration: how many candidates needed the exact estimate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vectordb_amd import _ffi
from vectordb_amd.builder import metric_id
from vectordb_amd.synthetic import build_tree_index_on_device

name = sys.argv[1] if len(sys.argv) > 1 else "rabitq10m"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
fanout, levels, leaf = cfg["tree"]
dti, q = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, dev, seed=0xC0DD1E, n_queries=cfg["n_queries"],
                                    metric=metric_id(cfg["metric"]), style=cfg.get("style", "sift"), code_kind=_ffi.CODE_RABITQ)
idx = dti.create(0)
for rf in (10, 40):
    idx.search_batch_device(q, cfg["k"], cfg["probes"], rf, 0)
    torch.cuda.synchronize()
    st = idx.last_stats()
    nq = q.shape[0]
    print(f"rf {rf}: candidates/query {st['vectors_scanned'] / nq:.0f}  selected/query {st['vectors_reranked'] / nq:.0f}  "
          f"exact estimates/query {st['rabitq_exact_estimates'] / nq:.0f}  exact-for-all queries {st['rabitq_exact_all_queries']} of {nq}")
