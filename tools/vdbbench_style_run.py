"""A run in the style of the harness the reference is benchmarked with
(docs/vectordbbench_integration/VECTORDBBENCH_INTEGRATION.md:160-182): load (insert + optimize), a serial search of
every test query for p95 / p99 latency and recall, and a batched pass for QPS, through the harness client
(vectordb_amd/vdbbench_client.py). The harness' datasets are downloads; this run takes synthetic rows of a case's
shape (labelled so) and computes the ground truth itself, exhaustively in f32 on the GPU.

    python tools/vdbbench_style_run.py --case Performance1536D50K [--probes 100 --rerank-factor 10]
"""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd.datasets import synthetic_mixture  # noqa: E402
from vectordb_amd.vdbbench_client import CuddleCaseConfig, CuddleVectorDB  # noqa: E402

CASES = {  # VECTORDBBENCH_INTEGRATION.md:401-407: vectors, dimensions, metric
    "Performance1536D50K": (50_000, 1536, "Cosine"),
    "Performance1536D500K": (500_000, 1536, "Cosine"),
    "Performance768D1M": (1_000_000, 768, "Cosine"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="Performance1536D50K", choices=sorted(CASES))
    ap.add_argument("--branching-factor", type=int, default=100)
    ap.add_argument("--target-leaf-size", type=int, default=100)
    ap.add_argument("--probes", type=int, default=100)
    ap.add_argument("--rerank-factor", type=int, default=10)
    ap.add_argument("--k", type=int, default=100)  # the harness' performance cases ask for 100 neighbours
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--path", default="/tmp/vectordb_bench.bin")
    a = ap.parse_args()
    n, dim, metric = CASES[a.case]
    dev = torch.device("cuda", 0)
    base, queries = synthetic_mixture(n, dim, seed=0xC0DD1E, device=dev, style="gist", n_queries=a.queries, latent_dim=64)
    bn = torch.nn.functional.normalize(base, dim=1)
    truth = torch.cat([(torch.nn.functional.normalize(queries[i:i + 256], dim=1) @ bn.T).topk(a.k, dim=1).indices
                       for i in range(0, a.queries, 256)]).cpu().numpy()
    del bn
    x, q = base.cpu().numpy(), queries.cpu().numpy()
    ids = np.random.default_rng(1).permutation(10 * n)[:n].astype(np.int64)  # harness ids are not array indices

    cfg = CuddleCaseConfig(a.branching_factor, a.target_leaf_size, a.probes, a.rerank_factor, metric)
    db = CuddleVectorDB(dim, {"path": a.path}, cfg, drop_old=True)
    t0 = time.perf_counter()
    with db.init():
        for i in range(0, n, 5000):
            cnt, err = db.insert_embeddings(x[i:i + 5000], ids[i:i + 5000])
            assert err is None and cnt == min(5000, n - i), err
        t_ins = time.perf_counter() - t0
        db.optimize()
    load_dur = time.perf_counter() - t0

    searcher = pickle.loads(pickle.dumps(db))  # the harness searches in a process of its own: configuration only
    t0 = time.perf_counter()
    with searcher.init():  # rebuilds from the staging file and the id mapping
        rebuild = time.perf_counter() - t0
        lat, hits = [], 0
        for i in range(a.queries):
            t1 = time.perf_counter()
            got = searcher.search_embedding(q[i], a.k)
            lat.append(time.perf_counter() - t1)
            hits += len(set(got) & set(ids[truth[i]].tolist()))
        searcher.search_embeddings(q, a.k)  # warm
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            batch = searcher.search_embeddings(q, a.k)
        t_batch = (time.perf_counter() - t1) / reps
        bhits = sum(len(set(batch[i]) & set(ids[truth[i]].tolist())) for i in range(a.queries))
    lat = np.sort(np.array(lat))
    print(json.dumps({
        "db": "cuddle-hip", "case": a.case, "data": "synthetic rows of the case's shape", "k": a.k,
        "load_dur_s": round(load_dur, 2), "insert_s": round(t_ins, 2), "subprocess_rebuild_s": round(rebuild, 2),
        "recall": round(hits / (a.queries * a.k), 4), "latency_p99_s": round(float(lat[int(0.99 * len(lat))]), 5),
        "latency_p95_s": round(float(lat[int(0.95 * len(lat))]), 5), "serial_qps": round(a.queries / float(lat.sum()), 1),
        "batched_qps": round(a.queries / t_batch, 1), "batched_recall": round(bhits / (a.queries * a.k), 4),
        "probes": a.probes, "rerank_factor": a.rerank_factor, "branching_factor": a.branching_factor,
        "target_leaf_size": a.target_leaf_size}))


if __name__ == "__main__":
    main()
