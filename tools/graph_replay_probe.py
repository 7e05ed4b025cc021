"""Does replaying one search as a captured HIP graph beat launching its kernels one by one?

    python tools/graph_replay_probe.py <config> <rerank_factor> [steps] [queries per call]

Builds the configuration's index as bench.py does, warms the handle up (work space allocated, routing
filters settled), captures one cuddle_search_batch_device call into a graph (relaxed capture mode: the
call's event query for the routing-filter feedback is left alone), and times `steps` direct calls
against `steps` replays on the same query batch, back to back and with a synchronisation after every call (the
latency a caller sees for one small request). Prints one JSON line. Development tool.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vectordb_amd import _ffi  # noqa: E402


def build(name, device):
    cfg = bench.CONFIGS[name]
    if "tree" in cfg:
        from vectordb_amd.builder import metric_id
        from vectordb_amd.synthetic import build_tree_index_on_device
        fanout, levels, leaf = cfg["tree"]
        dti, queries = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, device, seed=0xC0DD1E,
                                                  n_queries=cfg["n_queries"], metric=metric_id(cfg["metric"]),
                                                  style=cfg.get("style", "sift"),
                                                  code_kind=_ffi.CODE_RABITQ if cfg["code_kind"] == "rabitq" else _ffi.CODE_BINARY)
        return dti.create(0), queries, dti
    from vectordb_amd.builder import build_index
    from vectordb_amd.datasets import synthetic_mixture
    from vectordb_amd.index import DeviceIndex
    base, queries = synthetic_mixture(cfg["n"], cfg["dim"], seed=0xC0DD1E, device=device, style=cfg["style"],
                                      n_queries=cfg["n_queries"], latent_dim=cfg["latent_dim"])
    fi = build_index(base, cfg["branching"], cfg["leaf"], cfg["metric"], 20, seed=1, code_kind=cfg["code_kind"], threshold="mean")
    return DeviceIndex(fi, 0), queries, fi


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "gist1m"
    rf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    cfg = bench.CONFIGS[name]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx, queries, keep = build(name, device)
    if len(sys.argv) > 4:
        queries = queries[:int(sys.argv[4])].contiguous()
    nq, k, probes = queries.shape[0], cfg["k"], cfg["probes"]
    out = (torch.empty((nq, k), dtype=torch.int64, device=device), torch.empty((nq, k), dtype=torch.float32, device=device),
           torch.empty((nq,), dtype=torch.int32, device=device))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(40):  # past the routing-filter feedback's early snapshots
            idx.search_batch_device(queries, k, probes, rf, 0, *out)
        torch.cuda.synchronize()
        ref = [t.clone() for t in out]
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.search_batch_device(queries, k, probes, rf, 0, *out)
        torch.cuda.synchronize()
        direct = (time.perf_counter() - t0) / steps
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            idx.search_batch_device(queries, k, probes, rf, 0, *out, stream=torch.cuda.current_stream().cuda_stream)
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.search_batch_device(queries, k, probes, rf, 0, *out)
            torch.cuda.synchronize()
        direct_sync = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
            torch.cuda.synchronize()
        replay_sync = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "queries_per_call": nq, "rerank_factor": rf, "steps": steps, "direct_ms": direct * 1e3,
                      "graph_replay_ms": replay * 1e3, "direct_synced_ms": direct_sync * 1e3, "graph_replay_synced_ms": replay_sync * 1e3,
                      "replay_results_identical": same}))


if __name__ == "__main__":
    main()
