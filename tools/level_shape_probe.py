"""Level widths of a bench configuration's tree and the sizes of the per-query frontiers at each level (stats variant).
Development tool.   python tools/level_shape_probe.py <config>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
from graph_replay_probe import build  # noqa: E402


def main():
    name = sys.argv[1]
    cfg = bench.CONFIGS[name]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx, queries, keep = build(name, device)
    q = queries[:200].cpu().numpy()
    fi = keep if hasattr(keep, "child_offsets") else keep.to_flat()
    co = np.asarray(fi.child_offsets)
    nchild = co[1:] - co[:-1]
    level = np.full(len(nchild), -1)
    cur = np.asarray(fi.roots)
    l = 0
    while len(cur):
        level[cur] = l
        nxt = np.concatenate([np.asarray(fi.children)[co[n]:co[n + 1]] for n in cur]) if len(cur) else []
        print(f"level {l}: {len(cur)} nodes, {int((nchild[cur] == 0).sum())} leaves, children per inner node: "
              f"{nchild[cur][nchild[cur] > 0].mean() if (nchild[cur] > 0).any() else 0:.1f}")
        cur = np.asarray(nxt, dtype=np.int64)
        l += 1
    out = idx.search_batch(q, cfg["k"], cfg["probes"], cfg["rerank"] if "rerank" in cfg else 10, want_stats=True)
    print({k: v for k, v in out[3].items() if "level" in k or "probe" in k})


if __name__ == "__main__":
    main()
