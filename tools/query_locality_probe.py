"""Does the order of the queries in a batch matter? Times the same batch in its own order and sorted by the place (in leaf order) of each
query's nearest stored vector (neighbouring queries then reach the same leaves). Development tool.

    python tools/query_locality_probe.py <config> <rerank_factor>
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graph_replay_probe import build  # noqa: E402


def main():
    name, rf = sys.argv[1], int(sys.argv[2])
    cfg = bench.CONFIGS[name]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx, queries, keep = build(name, device)
    nq, k, probes = queries.shape[0], cfg["k"], cfg["probes"]
    out = (torch.empty((nq, k), dtype=torch.int64, device=device), torch.empty((nq, k), dtype=torch.float32, device=device),
           torch.empty((nq,), dtype=torch.int32, device=device))

    def run(q, steps=30):
        for _ in range(10):
            idx.search_batch_device(q, k, probes, rf, 0, *out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.search_batch_device(q, k, probes, rf, 0, *out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    a = run(queries)
    first = out[0][:, 0].clone()
    if hasattr(keep, "ids"):  # a host index: ids are the callers'; the place in leaf order is what neighbours share
        import numpy as np
        inv = np.empty(len(keep.ids), dtype=np.int64)
        inv[np.asarray(keep.ids, dtype=np.int64)] = np.arange(len(keep.ids))
        first = torch.from_numpy(inv).to(device)[first]
    order = torch.argsort(first)
    qs = queries[order].contiguous()
    b = run(qs)
    print(json.dumps({"config": name, "rerank_factor": rf, "ms_in_batch_order": a, "ms_sorted_by_nearest_id": b}))


if __name__ == "__main__":
    main()
