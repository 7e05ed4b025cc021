#!/usr/bin/env python3
"""Randomised parity sweep (development aid): random data shapes, tree shapes, metrics and search
parameters; the GPU search must return the CPU oracle's ids and distance bits for every query.
  python tools/random_parity_sweep.py [--cases 30] [--seed 1] [--boundary]
--boundary draws the shapes and parameters from the edges instead (dimension 1, one-vector leaves, probes and
k * rerank_factor at and around every capacity of the kernels); a search the library refuses with
CUDDLE_ERR_UNSUPPORTED is reported but is not a mismatch, any other error is."""
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


def random_tree_index(rng, d, metric, kind):
    """An irregular forest made by hand (no k-means): 1..12 roots, depth up to 6, fan-out 1..9, leaves at any depth
    holding 0, 1, 2, 3, 63, 64, 65, 200 or 1500 vectors; centroids near their parent's (they need not be means for
    the search to be defined). Returns (vectors in stored order, FlatIndex)."""
    import torch
    from vectordb_amd.builder import make_rotation, metric_id, rabitq_encode
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    cents, kids, sizes = [], [], []

    def add(parent_c, depth):
        c = (parent_c + 0.5 * rng.standard_normal(d)).astype(np.float32)
        me = len(cents)
        cents.append(c); kids.append([]); sizes.append(0)
        if depth >= int(rng.integers(1, 7)) or len(cents) > 3000:
            sizes[me] = int(rng.choice([0, 1, 2, 3, 63, 64, 65, 200, 1500], p=[.1, .15, .15, .1, .1, .1, .1, .15, .05]))
        else:
            for _ in range(int(rng.integers(1, 10))):
                kids[me].append(add(c, depth + 1))
        return me

    roots = [add(np.zeros(d, dtype=np.float32), 1) for _ in range(int(rng.integers(1, 13)))]
    n_nodes = len(cents)
    child_offsets = np.zeros(n_nodes + 1, dtype=np.uint32)
    vec_offsets = np.zeros(n_nodes + 1, dtype=np.uint64)
    children, rows = [], []
    for i in range(n_nodes):
        children.extend(kids[i])
        child_offsets[i + 1] = len(children)
        vec_offsets[i + 1] = vec_offsets[i] + np.uint64(sizes[i])
        if sizes[i]:
            rows.append((cents[i][None, :] + 0.2 * rng.standard_normal((sizes[i], d))).astype(np.float32))
    x = np.concatenate(rows) if rows else np.zeros((0, d), dtype=np.float32)
    if rng.random() < 0.3:
        x = np.round(x * 2) / 2
    fi = FlatIndex(dim=d, metric=metric_id(metric), code_kind=_ffi.CODE_BINARY if kind == "binary" else _ffi.CODE_RABITQ,
                   centroids=np.stack(cents), child_offsets=child_offsets, children=np.asarray(children, dtype=np.uint32),
                   roots=np.asarray(roots, dtype=np.uint32), vec_offsets=vec_offsets,
                   ids=rng.permutation(x.shape[0]).astype(np.uint64), vectors=x)
    if kind == "binary":
        fi.threshold = float(x.mean()) if x.size else 0.0
        fi.codes = pack_binary_codes(x, fi.threshold)
    else:
        fi.rabitq_rot_t = make_rotation(d, int(rng.integers(0, 1000)))
        fi.rabitq_codes, fi.rabitq_norm, fi.rabitq_ip_oo = rabitq_encode(torch.from_numpy(x), torch.from_numpy(fi.rabitq_rot_t))
    return x, fi.normalised()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--boundary", action="store_true")
    ap.add_argument("--trees", action="store_true", help="hand-made irregular forests instead of k-means trees")
    ap.add_argument("--sharded", action="store_true",
                    help="also run the exact leaf-sharded protocol over 2..5 shards of the index (separate handles on the "
                         "one GPU) and compare it with the unsharded search bit for bit")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        if args.boundary:
            d = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 129, 254, 255, 256, 257, 511, 513, 1000]))
            n = int(rng.choice([1, 2, 7, 300, 3000, 20000]))
            metric = str(rng.choice(["l2", "cosine", "dot"]))
            branching = int(rng.choice([2, 3, 8, 64, 300]))
            leaf = int(rng.choice([10, 11, 40, 500]))  # the builder keeps the reference's minimum of 10
            kind = str(rng.choice(["binary", "binary", "rabitq"]))
            probes = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 2047, 2048, 4095, 4096]))
            k = int(rng.choice([1, 2, 9, 10, 11, 64, 65, 100, 500, 1000, 3000]))
            rf = int(rng.choice([1, 2, 7, 100, 1000]))
            flags = int(rng.choice([0, 0, _ffi.SEARCH_STATS_VARIANT, _ffi.SEARCH_ROUTE_GROUPED, _ffi.SEARCH_ROUTE_EXACT_ONLY]))
        else:
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
        t0 = time.time()
        if args.trees:
            d = min(d, 257)
            x, fi = random_tree_index(rng, d, metric, kind)
            n = x.shape[0]
            if n == 0:
                continue
        else:
            x = clustered(n, d, int(rng.integers(1 if args.boundary else 5, 200)), int(rng.integers(0, 1 << 30)),
                          sigma=float(rng.choice([0.1, 0.3, 1.0])))
            if rng.random() < 0.3:
                x = np.round(x * 2) / 2  # coarse values: ties everywhere
            fi = build_index(x, branching, leaf, metric, 5, seed=int(rng.integers(0, 1000)), code_kind=kind)
        dev = DeviceIndex(fi, 0)
        oi = orc.OracleIndex(fi)
        nq = int(rng.choice([1, 3, 64, 65])) if args.boundary else 64
        q = queries_like(x, nq, int(rng.integers(0, 1000)), noise=float(rng.choice([0.0, 0.05, 0.5])))
        desc = (f"case {case}: n={n} d={d} {metric} b={branching} leaf={leaf} {kind} probes={probes} k={k} rf={rf} flags={flags} "
                f"nq={nq} levels={fi.n_levels()} nodes={fi.n_nodes}")
        try:
            ids, dist, cnt, st = dev.search_batch(q, k, probes, rf, flags, want_stats=True)
        except _ffi.CuddleError as e:
            if e.status == _ffi.ERR_UNSUPPORTED:
                print(desc, "UNSUPPORTED:", e, flush=True)
                continue
            print(desc, "ERROR:", e, flush=True)
            bad += 1
            continue
        rc, oids, odist, ocnt = oi.search_batch(q, k, probes, rf, flags)
        ok = rc == 0 and np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
        if kind == "binary":
            ok = ok and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
        else:
            ok = ok and np.allclose(dist, odist, rtol=1e-5, atol=0, equal_nan=True)
        if rc == orc.ORC_ERR_NAN:
            print(desc, "NaN distances (the reference panics on them): not compared", flush=True)
            continue
        if not ok:
            diff = np.argwhere(ids != oids)
            print(f"  oracle rc={rc}; counts equal: {np.array_equal(cnt, ocnt)}; first id difference at {diff[0] if len(diff) else None}; "
                  f"NaNs gpu/oracle: {int(np.isnan(dist).sum())}/{int(np.isnan(odist).sum())}", flush=True)
            if len(diff):
                qi, j = diff[0]
                print("  gpu   ", ids[qi, max(0, j - 2):j + 3], dist[qi, max(0, j - 2):j + 3], flush=True)
                print("  oracle", oids[qi, max(0, j - 2):j + 3], odist[qi, max(0, j - 2):j + 3], flush=True)
        bad += 0 if ok else 1
        shard_note = ""
        if args.sharded and ok and not (flags & _ffi.SEARCH_STATS_VARIANT):
            import torch
            from vectordb_amd.sharded import exact_search_in_process
            world = int(rng.choice([2, 3, 5]))
            if world * k <= 1024:
                shards = [DeviceIndex(fi.shard(r, world), 0) for r in range(world)]
                dq = torch.from_numpy(q).cuda(0)
                try:
                    sids, sdist, scnt = exact_search_in_process(shards, dq, k, probes, rf, shard_routing=bool(rng.integers(0, 2)),
                                                                packed=bool(rng.integers(0, 2)))
                    torch.cuda.synchronize()
                    same = (np.array_equal(sids.cpu().numpy().astype(np.uint64), ids) and np.array_equal(scnt.cpu().numpy(), cnt)
                            and np.array_equal(sdist.cpu().numpy().view(np.uint32), dist.view(np.uint32)))
                    shard_note = f" sharded x{world}: {'ok' if same else 'MISMATCH'}"
                    bad += 0 if same else 1
                except _ffi.CuddleError as e:
                    shard_note = f" sharded x{world}: {'UNSUPPORTED' if e.status == _ffi.ERR_UNSUPPORTED else 'ERROR'}: {e}"
                    bad += 0 if e.status == _ffi.ERR_UNSUPPORTED else 1
                for sh in shards:
                    sh.close()
        print(desc, f"fallbacks={st['route_fallbacks']} {'ok' if ok else 'MISMATCH'}{shard_note} ({time.time() - t0:.1f}s)", flush=True)
        dev.close()
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
