#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: small seeded indexes + queries + the CPU oracle's outputs for
every stage (leaf lists, candidates, phase-1 selection, reranked distances, final top-k).

The reference ships no fixtures for this path and cannot be built or run here (Rust, aarch64-only),
so the expected values come from oracle/cuddle_oracle.c, which is pinned by the reference's own
analytic known-answer tests (tests/test_oracle_kats.py). The fixtures freeze those outputs so that
an accidental change of the oracle or of the kernels shows up as a diff against committed data.

    python tests/golden/make_golden.py [case names: default all]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import clustered, queries_like  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from vectordb_amd.builder import build_index  # noqa: E402

CASES = {
    # name: (n, d, clusters, branching, leaf, metric, code_kind, integer, k, probes, rf, flags)
    "binary_l2_d72": (900, 72, 12, 6, 20, "l2", "binary", True, 5, 3, 4, 0),
    "binary_cosine_d33": (700, 33, 10, 5, 20, "cosine", "binary", False, 7, 4, 3, 0),
    "binary_dot_stats_d64": (800, 64, 10, 8, 25, "dot", "binary", False, 4, 2, 5, 1),
    "rabitq_l2_d40": (600, 40, 8, 5, 20, "l2", "rabitq", False, 5, 3, 4, 0),
    # the shapes of the benchmark configurations, at fixture size: 960-d L2 k=10 probes 200 (15-word codes),
    # 768-d cosine RaBitQ k=100 probes 100 rerank 10, 128-d L2 k=10 probes 100 (the leaf-sharded configs)
    "gist_shape_l2_d960": (800, 960, 10, 10, 10, "l2", "binary", True, 10, 200, 10, 0),
    "rabitq_cosine_d768_k100": (400, 768, 8, 6, 12, "cosine", "rabitq", False, 100, 100, 10, 0),
    "shard_shape_l2_d128": (2400, 128, 24, 8, 12, "l2", "binary", True, 10, 100, 25, 0),
}
FIELDS = ("centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes", "rabitq_rot_t",
          "rabitq_codes", "rabitq_norm", "rabitq_ip_oo")


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (n, d, nc, br, leaf, metric, kind, integer, k, probes, rf, flags) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        x = clustered(n, d, nc, seed=len(name), integer=integer)
        if d >= 512 and not integer:
            x = (np.round(x * 128) / 128).astype(np.float32)  # wide float rows: values on a grid compress to a small file
        fi = build_index(x, br, leaf, metric, 8, seed=7, code_kind=kind)
        q = queries_like(x, 12, seed=3, noise=2.0 if integer else 0.05)
        q[0] = x[0]  # an indexed vector as query: distance 0 to itself for L2
        oi = orc.OracleIndex(fi)
        rec = {"dim": d, "metric": fi.metric, "code_kind": fi.code_kind, "threshold": np.float32(fi.threshold),
               "queries": q, "k": k, "probes": probes, "rf": rf, "flags": flags}
        for f in FIELDS:
            a = getattr(fi, f)
            if a is not None:
                rec["index_" + f] = a
        per = {key: [] for key in ("leaves", "cand_pos", "cand_key", "sel_pos", "sel_key", "sel_dist", "ids", "dist", "count")}
        for i in range(q.shape[0]):
            r = oi.search(q[i], k, probes, rf, flags, trace=True)
            assert r["rc"] == 0, (name, i, r["rc"])
            for key in ("leaves", "cand_pos", "cand_key", "sel_pos", "sel_key", "sel_dist"):
                per[key].append(r[key])
            per["ids"].append(r["ids"])
            per["dist"].append(r["dist"])
            per["count"].append(r["count"])
        for key in ("leaves", "cand_pos", "cand_key", "sel_pos", "sel_key", "sel_dist"):
            rec["exp_" + key + "_len"] = np.array([len(v) for v in per[key]], dtype=np.int64)
            rec["exp_" + key] = np.concatenate(per[key]) if per[key] else np.zeros(0)
        rec["exp_ids"] = np.stack(per["ids"])
        rec["exp_dist"] = np.stack(per["dist"])
        rec["exp_count"] = np.array(per["count"], dtype=np.uint32)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **rec)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
