#!/usr/bin/env python3
"""Times cuddle_kmeans_assign (KMeans::assign, kmeans.rs:136-160) on synthetic clustered data, with the
plain torch matmul + argmin of the same shape and the CPU oracle (on a sample) beside it.
  python tools/bench_kmeans_assign.py [--n 1000000] [--k 10000] [--dim 128] [--metric l2] [--steps 5]
Prints one JSON line per run."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd.builder import kmeans_assign, metric_id, _pairwise  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=10000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=2000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    centres = torch.randn((max(args.k // 10, 1), args.dim), generator=g, device=dev)
    x = centres[torch.randint(0, centres.shape[0], (args.n,), generator=g, device=dev)] + 0.3 * torch.randn((args.n, args.dim), generator=g, device=dev)
    c = x[torch.randint(0, args.n, (args.k,), generator=g, device=dev)].clone()
    m = metric_id(args.metric)
    a, d, fb = kmeans_assign(x, c, m)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, d, fb = kmeans_assign(x, c, m)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # plain torch of the same shape (approximate arithmetic, materialises chunk x k distances)
    def torch_assign():
        out = torch.empty(args.n, dtype=torch.long, device=dev)
        chunk = max(1, min(args.n, (1 << 28) // max(args.k, 1)))
        for s in range(0, args.n, chunk):
            out[s:s + chunk] = _pairwise(x[s:s + chunk], c, m).argmin(dim=1)
        return out
    ta = torch_assign()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = torch_assign()
    torch.cuda.synchronize()
    torch_ms = (time.perf_counter() - t0) / args.steps * 1e3
    out = {"op": "kmeans_assign", "n": args.n, "k": args.k, "dim": args.dim, "metric": args.metric,
           "ms": ms, "useful_TFLOPs": 2.0 * args.n * args.k * args.dim / (ms * 1e-3) / 1e12,
           "vectors_per_s": args.n / (ms * 1e-3), "exact_fallback_rows": fb,
           "torch_matmul_argmin_ms": torch_ms, "agree_with_torch": float((ta.int() == a).float().mean().item())}
    if args.cpu_sample:
        from oracle import oracle as orc
        s = min(args.cpu_sample, args.n)
        xs, cs = x[:s].cpu().numpy(), c.cpu().numpy()
        t0 = time.perf_counter()
        ea, ed = orc.kmeans_assign(xs, cs, m)
        cpu_s = time.perf_counter() - t0
        out["cpu_oracle_vectors_per_s"] = s / cpu_s
        out["cpu_threads"] = orc.num_threads() if hasattr(orc, "num_threads") else None
        out["identical_to_oracle_on_sample"] = bool(np.array_equal(ea, a[:s].cpu().numpy().astype(np.uint32)) and
                                                    np.array_equal(ed.view(np.uint32), d[:s].cpu().numpy().view(np.uint32)))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
