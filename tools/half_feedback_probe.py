"""Fallbacks of the routing filter over the first calls on a fresh handle of a by-construction configuration: shows the
failure feedback returning a level to the split filter (gist10m: 184, then 0).
usage: python tools/half_feedback_probe.py <config>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from vectordb_amd import _ffi
from vectordb_amd.builder import metric_id
from vectordb_amd.synthetic import build_tree_index_on_device
name = sys.argv[1]
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
fanout, levels, leaf = cfg["tree"]
dti, q = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, dev, seed=0xC0DD1E, n_queries=cfg["n_queries"], metric=metric_id(cfg["metric"]), style=cfg.get("style", "sift"), code_kind=_ffi.CODE_RABITQ if cfg["code_kind"] == "rabitq" else _ffi.CODE_BINARY)
idx = dti.create(0)
qn = q.cpu().numpy()
for it in range(6):
    ids, dist, cnt, st = idx.search_batch(qn, cfg["k"], cfg["probes"], 10, want_stats=True)
    print(name, "call", it, "route_fallbacks", st["route_fallbacks"], flush=True)
