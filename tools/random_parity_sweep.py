#!/usr/bin/env python3
"""Randomised parity sweep (development aid): random data shapes, tree shapes, metrics and search
parameters; the GPU search must return the CPU oracle's ids and distance bits for every query.
  python tools/random_parity_sweep.py [--cases 30] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import clustered, queries_like  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from vectordb_amd import _ffi  # noqa: E402
from vectordb_amd.builder import build_index  # noqa: E402
from vectordb_amd.index import DeviceIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        d = int(rng.choice([8, 24, 37, 64, 96, 128, 200, 256, 300]))
        n = int(rng.choice([8000, 20000, 50000]))
        metric = str(rng.choice(["l2", "cosine", "dot"]))
        branching = int(rng.choice([8, 20, 50, 100]))
        leaf = int(rng.choice([20, 40, 100]))
        kind = str(rng.choice(["binary", "binary", "rabitq"]))
        probes = int(rng.choice([3, 10, 40, 100, 300]))
        k = int(rng.choice([1, 10, 50]))
        rf = int(rng.choice([1, 5, 30]))
        flags = int(rng.choice([0, 0, _ffi.SEARCH_STATS_VARIANT, _ffi.SEARCH_ROUTE_GROUPED]))
        x = clustered(n, d, int(rng.integers(5, 200)), int(rng.integers(0, 1 << 30)), sigma=float(rng.choice([0.1, 0.3, 1.0])))
        if rng.random() < 0.3:
            x = np.round(x * 2) / 2  # coarse values: ties everywhere
        t0 = time.time()
        fi = build_index(x, branching, leaf, metric, 5, seed=int(rng.integers(0, 1000)), code_kind=kind)
        dev = DeviceIndex(fi, 0)
        oi = orc.OracleIndex(fi)
        q = queries_like(x, 64, int(rng.integers(0, 1000)), noise=float(rng.choice([0.0, 0.05, 0.5])))
        ids, dist, cnt, st = dev.search_batch(q, k, probes, rf, flags, want_stats=True)
        rc, oids, odist, ocnt = oi.search_batch(q, k, probes, rf, flags)
        ok = rc == 0 and np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
        if kind == "binary":
            ok = ok and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
        else:
            ok = ok and np.allclose(dist, odist, rtol=1e-5, atol=0, equal_nan=True)
        bad += 0 if ok else 1
        print(f"case {case}: n={n} d={d} {metric} b={branching} leaf={leaf} {kind} probes={probes} k={k} rf={rf} flags={flags} "
              f"levels={fi.n_levels()} fallbacks={st['route_fallbacks']} {'ok' if ok else 'MISMATCH'} ({time.time() - t0:.1f}s)", flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
