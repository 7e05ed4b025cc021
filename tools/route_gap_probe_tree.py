"""route_gap_probe.py for the by-construction trees (shard12m / shard125m / gist10m / rabitq10m shapes): the level
whose frontiers differ per query. Per query: the children of the probes nearest nodes of the level above; the gap
between the probes-th and the (probes+m)-th distance among them, in units of the filter's error scale - with the
level-wide centring the library uses, and with every family centred on its parent.
usage: python tools/route_gap_probe_tree.py <config>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vectordb_amd import _ffi  # noqa: E402
from vectordb_amd.builder import metric_id  # noqa: E402
from vectordb_amd.synthetic import build_tree_index_on_device  # noqa: E402

name = sys.argv[1]
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
fanout, levels, leaf = cfg["tree"]
leaf = min(leaf, 4)  # the vectors do not matter here
dti, q = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, dev, seed=0xC0DD1E, n_queries=256,
                                    metric=metric_id(cfg["metric"]), style=cfg.get("style", "sift"), code_kind=_ffi.CODE_BINARY)
P = cfg["probes"]
cent = dti.centroids.double()
q = q.double()
n0, n1 = fanout, fanout * fanout
lvl1 = cent[n0:n0 + n1]
lvl2 = cent[n0 + n1:n0 + n1 + n1 * fanout]
mu2 = lvl2.mean(0)
max_c2 = ((lvl2 - mu2) ** 2).sum(1).max()
d1 = torch.cdist(q, lvl1) ** 2
top = d1.topk(P, dim=1, largest=False).indices  # [nq, P] parents
res = {m: ([], []) for m in (8, 16, 24, 32, 50, 100)}
for i in range(q.shape[0]):
    par = top[i]
    kids = (par[:, None] * fanout + torch.arange(fanout, device=dev)[None, :]).reshape(-1)
    c = lvl2[kids]
    d = ((c - q[i]) ** 2).sum(1)
    ds, _ = d.sort()
    scale_level = ((q[i] - mu2) ** 2).sum() + max_c2
    pc = lvl1[par]                                     # parents' centroids
    qp = ((q[i][None, :] - pc) ** 2).sum(1)            # |q - mu_p|^2 per parent
    cp = ((c.reshape(P, fanout, -1) - pc[:, None, :]) ** 2).sum(2).max(1).values
    scale_parent = (qp + cp).max()
    for m in res:
        if P + m <= ds.shape[0]:
            g = (ds[P + m - 1] - ds[P - 1])
            res[m][0].append(float(g / scale_level)); res[m][1].append(float(g / scale_parent))
D = cfg["dim"]
print(f"{name}: dim {D}, probes {P}, frontier {P * fanout}")
for m, (a, b) in res.items():
    a, b = np.array(a), np.array(b)
    print(f"m={m:4d}: level-centred gap/scale min {a.min():.3e} p10 {np.percentile(a, 10):.3e} med {np.median(a):.3e} | "
          f"parent-centred min {b.min():.3e} p10 {np.percentile(b, 10):.3e} med {np.median(b):.3e}")
print("need gap/scale > 2 r:  split bf16 r =", (2.5 * D + 64) * 2.0 ** -24 + 3.25 * 2.0 ** -16, " half r =", (2.5 * D + 64) * 2.0 ** -24 + 1.01 * 2.0 ** -10)
