import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vectordb_amd.datasets import synthetic_mixture
from vectordb_amd.pyvectordb import PyVectorDB
dev = torch.device("cuda", 0)
base, queries = synthetic_mixture(50000, 1536, seed=1, device=dev, style="gist", n_queries=200, latent_dim=64)
db = PyVectorDB(1536, metric="Cosine", temp_file_path="/tmp/latprobe.bin")
db.insert_embeddings(base.cpu().numpy()); db.optimize()
q = queries.cpu().numpy()
ix = db.index
for _ in range(20): db.search(q[0], 100, 100, 10)
def t(f, n=200):
    t0 = time.perf_counter()
    for i in range(n): f(i)
    return (time.perf_counter() - t0) / n * 1e3
print("db.search list in        ", t(lambda i: db.search(q[i].tolist(), 100, 100, 10)))
print("db.search ndarray in     ", t(lambda i: db.search(q[i], 100, 100, 10)))
print("index.search_batch nq=1  ", t(lambda i: ix.search_batch(q[i:i+1], 100, 100, 10)))
d = ix.dev
out = (torch.empty((1, 100), dtype=torch.int64, device=dev), torch.empty((1, 100), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
def devcall(i):
    d.search_batch_device(queries[i:i+1], 100, 100, 10, 0, *out); torch.cuda.synchronize()
print("search_batch_device+sync ", t(devcall))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200): db.search(q[i], 100, 100, 10)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
