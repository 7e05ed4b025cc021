"""How much room does the routing proof have? For the widest level of a benchmark index: per query the gap between
the exact probes-th distance and the (probes+m)-th, in units of (|q-mu|^2 + max|c-mu|^2), the quantity every
filter error bound of the routing contraction is proportional to. A filter with relative error bound r passes
the proof with m extra candidates when gap_m > 2 r.
usage: python tools/route_gap_probe.py [gist1m|sift1m]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vectordb_amd.builder import build_index  # noqa: E402
from vectordb_amd.datasets import synthetic_mixture  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "gist1m"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
base, q = synthetic_mixture(cfg["n"], cfg["dim"], seed=0xC0DD1E, device=dev, style=cfg["style"], n_queries=1000,
                            latent_dim=cfg["latent_dim"], query_block=1000)
fi = build_index(base, cfg["branching"], cfg["leaf"], cfg["metric"], 20, seed=1, code_kind="binary")
del base
# widest level, breadth first
cur = list(fi.roots)
best = cur
while cur:
    nxt = []
    for v in cur:
        nxt.extend(fi.children[fi.child_offsets[v]:fi.child_offsets[v + 1]])
    if len(nxt) > len(best):
        best = nxt
    cur = nxt
c = torch.from_numpy(fi.centroids[np.asarray(best, dtype=np.int64)]).to(dev).double()
qd = q.double()
mu = c.mean(0)
cc, qq = c - mu, qd - mu
d = (qq * qq).sum(1)[:, None] + (cc * cc).sum(1)[None, :] - 2 * qq @ cc.T
ds, _ = d.sort(dim=1)
P = cfg["probes"]
scale = (qq * qq).sum(1) + (cc * cc).sum(1).max()
print(f"{name}: level of {len(best)} nodes, probes {P}, dim {cfg['dim']}")
for m in (8, 16, 25, 32, 50, 64, 100, 200):
    if P + m > ds.shape[1]:
        break
    gap = ((ds[:, P + m - 1] - ds[:, P - 1]) / scale).cpu().numpy()
    print(f"m={m:4d}: gap/scale min {gap.min():.3e}  p1 {np.percentile(gap, 1):.3e}  p10 {np.percentile(gap, 10):.3e}  median {np.median(gap):.3e}")
D = cfg["dim"]
print("bound r: split bf16", (2.5 * D + 64) * 2.0 ** -24 + 3.25 * 2.0 ** -16, " single fp16 ~", (2.5 * D + 64) * 2.0 ** -24 + 2.0 ** -10 * 1.01)
