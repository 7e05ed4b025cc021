#!/usr/bin/env python3
"""One rank's share of the exact leaf-sharded step, timed on one GPU without the collectives: route
nq/world queries, scan this shard's leaves for the whole batch, ties, finish, merge of `world`
messages. A proxy for the N-GPU step time minus communication (the histogram used for the threshold
is this shard's own, so the selection is not the global one; the work per kernel is the same).
  python tools/shard_step_time.py [--world 8] [--config gist1m] [--steps 50]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vectordb_amd.builder import build_index  # noqa: E402
from vectordb_amd.datasets import synthetic_mixture  # noqa: E402
from vectordb_amd.index import DeviceIndex  # noqa: E402
from vectordb_amd.sharded import merge_gathered_packed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", default="gist1m")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--rerank", type=int, default=10)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    x, q = synthetic_mixture(cfg["n"], cfg["dim"], seed=0xC0DD1E, device=dev, style=cfg["style"], n_queries=cfg["n_queries"],
                             latent_dim=cfg["latent_dim"])
    fi = build_index(x, cfg["branching"], cfg["leaf"], cfg["metric"], 20, seed=1, code_kind=cfg["code_kind"], threshold="mean")
    del x
    torch.cuda.empty_cache()
    k, probes, rf, world = cfg["k"], cfg["probes"], args.rerank, args.world
    shard = DeviceIndex(fi.shard(0, world), 0)
    nq = q.shape[0]
    per = (nq + world - 1) // world
    stride = shard.leaf_stride(probes)
    full_l, full_c = shard.route_device(q, probes)
    words = shard.packed_len(nq, k)

    def step():
        mine = torch.empty((per, stride + 1), dtype=torch.int32, device=dev)
        _, c = shard.route_device(q[:per], probes, out=mine)
        mine[:per, stride] = c
        hist = shard.shard_scan(q, k, probes, rf, leaves=full_l, leaf_count=full_c)
        ties = shard.shard_ties(hist)
        everyone = torch.empty((world, words), dtype=torch.int64, device=dev)
        shard.shard_finish(q, ties, packed=everyone[0])
        return merge_gathered_packed(everyone[:1].expand(world, words).contiguous(), nq, k)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(json.dumps({"config": args.config, "world": world, "ms_per_step": t_all / args.steps * 1e3,
                      "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
                      "implied_qps_without_communication": nq / (t_all / args.steps)}))


if __name__ == "__main__":
    main()
