#!/usr/bin/env python3
"""Benchmark of the batched two-phase ANN query path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU. Either the launcher provides the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when RANK is not set, this script
starts the N rank processes itself (before anything touches a GPU) and relays rank 0's line.

One "step" = one pass of the hot path (route -> code scan + select -> fp32 rerank -> top-k) over
one batch of queries already resident in HBM; the timed loop rotates through 8 different query
batches so that no step replays the one before it. Default workload: "GIST1M (1M x 960-d f32, L2,
k=10) two-phase search on one MI355X" — GIST1M itself is not in the container and cannot be
fetched, so the data is a seeded GIST1M-shaped synthetic mixture, labelled as such. probes = 200
and the rerank sweep {5,10,25,50,100} are the reference's GIST settings
(/root/reference/benches/gist_benchmark.rs:177-183); the timed operating point is the cheapest
sweep point whose recall@10 against brute-force ground truth is >= 0.95.

Rank 0 prints ONE JSON line: metric/value/unit (whole-job queries per second on the default
workload), the timing contract fields, recall, batch-latency percentiles, `roofline` (dominant
kernel = the code scan: algorithmic code bytes per launch / mean launch time from HIP events on the
launch stream, against the 8 TB/s HBM peak; which scan ran - per query or leaf by leaf - is
`scan_mode`, and `ceiling_GBps` is the rate of the per-query scan's own loads with nothing behind
them, timed in the same run), `cpu_baseline` (the CPU oracle timed on this host, and whether the GPU's
ids and distance bits equal its results) and, at N = 1 on the default workload, `configs`: the same
measurement for every other configuration of BASELINE.json (gist10m = the default's shape on a code
array that cannot sit in the 256 MiB Infinity Cache, also reported as `roofline_hbm_resident`;
sift1m; rabitq10m; shard12m; shard125m), each with its own recall sweep, stage times, roofline block
and full-size oracle check, within CUDDLE_BENCH_BUDGET_S seconds (default 400; what does not fit is
named in `configs_skipped`). N > 1 adds `comm_ms`: every collective of the sharded protocol by name.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
N_BATCHES = 8           # distinct query batches the timed loop rotates through
STAGE_SAMPLE = 4        # per-stage HIP events are recorded on every STAGE_SAMPLE-th search of the timed loop

CONFIGS = {
    # name: data shape + the reference's search settings for that dataset
    "gist1m": dict(n=1_000_000, dim=960, n_queries=1000, k=10, metric="l2", style="gist", probes=200,
                   rerank_sweep=[5, 10, 25, 50, 100, 200, 400, 1000], branching=100, leaf=100, code_kind="binary",
                   latent_dim=48,
                   label="GIST1M-shaped synthetic (1M x 960 f32: two-level Gaussian mixture of intrinsic dimension 48 "
                         "embedded linearly in 960-d)"),
    "sift1m": dict(n=1_000_000, dim=128, n_queries=10_000, k=10, metric="l2", style="sift", probes=100,
                   rerank_sweep=[25, 75, 150, 300, 600, 1000], branching=100, leaf=100, code_kind="binary",
                   latent_dim=24,
                   label="SIFT1M-shaped synthetic (1M x 128 integer-valued f32: two-level Gaussian mixture of intrinsic "
                         "dimension 24 embedded linearly in 128-d)"),
    # the GIST shape past the Infinity Cache: 10M x 960 (1.2 GB of codes, 38 GB of fp32 rows), tree by construction
    "gist10m": dict(n=46 ** 3 * 103, dim=960, n_queries=1000, k=10, metric="l2", style="gauss", probes=200,
                    rerank_sweep=[5, 10, 25, 50, 100], code_kind="binary", tree=(46, 3, 103), gt_queries=500,
                    label="synthetic 10M x 960 f32 hierarchical Gaussian (1.2 GB of binary codes + 38 GB of fp32 rows in HBM), "
                          "tree by construction (46-way x 3 levels, 103-vector leaves)"),
    # one GPU's share of the leaf-sharded 100M / 1B x 128-d configs (index by construction, built on the device).
    # With N ranks the index is N such shards (N x 12.5M / N x 125M vectors): every rank generates its own leaves.
    "shard12m": dict(n=12_500_000, dim=128, n_queries=4096, k=10, metric="l2", style="sift", probes=100,
                     rerank_sweep=[5, 10, 25, 50, 100, 200], code_kind="binary", tree=(50, 3, 100), gt_queries=1000,
                     label="12.5M x 128 integer-valued f32 per GPU (one of 8 leaf shards of a synthetic 100M x 128 index), "
                           "hierarchical Gaussian, tree by construction (50-way x 3 levels, 100-vector leaves)"),
    "shard125m": dict(n=125_000_000, dim=128, n_queries=4096, k=10, metric="l2", style="sift", probes=100,
                      rerank_sweep=[5, 10, 25, 50, 100, 200], code_kind="binary", tree=(100, 3, 125), gt_queries=1000,
                      label="125M x 128 integer-valued f32 per GPU (one of 8 leaf shards of a synthetic 1B x 128 index: "
                            "2 GB codes + 64 GB fp32 in HBM), tree by construction (100-way x 3 levels, 125-vector leaves)"),
    # config 3: RaBitQ codes, cosine, k = 100 (tree by construction: 46-way x 3 levels, 103-vector leaves = 10.03M vectors)
    "rabitq10m": dict(n=46 ** 3 * 103, dim=768, n_queries=4096, k=100, metric="cosine", style="gauss", probes=100,
                      rerank_sweep=[10, 20, 40], code_kind="rabitq", tree=(46, 3, 103), gt_queries=500,
                      label="synthetic 10M x 768 f32 hierarchical Gaussian, RaBitQ codes (96 B + 8 B per vector), cosine, "
                            "tree by construction (46-way x 3 levels, 103-vector leaves)"),
    "tiny": dict(n=20_000, dim=64, n_queries=256, k=10, metric="l2", style="gist", probes=8,
                 rerank_sweep=[5, 10, 25, 50, 100], branching=16, leaf=50, code_kind="binary", latent_dim=0,
                 label="tiny synthetic (20k x 64)"),
    "tinytree": dict(n=8 ** 3 * 20, dim=32, n_queries=128, k=10, metric="l2", style="sift", probes=16,
                     rerank_sweep=[5, 10, 25, 50], code_kind="binary", tree=(8, 3, 20), gt_queries=128,
                     label="tiny tree by construction (8-way x 3 levels, 20-vector leaves per GPU)"),
    "tinyrq": dict(n=8 ** 3 * 20, dim=48, n_queries=128, k=10, metric="cosine", style="gauss", probes=16,
                   rerank_sweep=[5, 10, 25], code_kind="rabitq", tree=(8, 3, 20), gt_queries=128,
                   label="tiny RaBitQ tree by construction (8-way x 3 levels, 20-vector leaves per GPU)"),
}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def trace(msg):
    """Progress of every rank (CUDDLE_BENCH_TRACE=1): where a multi-rank run stops."""
    if os.environ.get("CUDDLE_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n: int) -> int:
    """Start the N rank processes of this same command line (fresh interpreters: this parent never touches a
    GPU), wait for them, relay rank 0's output. A rank that fails ends the others and the job."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    out0.seek(0)
    for line in out0:  # rank 0's result line; anything else a library printed on stdout goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    return rc


def broadcast_flat(fi, src_rank, device):
    """Rank `src_rank` sends its FlatIndex to every rank (broadcast of each array)."""
    import torch
    import torch.distributed as dist

    from vectordb_amd.flat_index import FlatIndex

    rank = dist.get_rank()
    names = ["centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes"]
    meta = [None]
    if rank == src_rank:
        meta[0] = dict(dim=fi.dim, metric=fi.metric, code_kind=fi.code_kind, threshold=fi.threshold,
                       arrays={n: (getattr(fi, n).shape, str(getattr(fi, n).dtype)) for n in names})
    dist.broadcast_object_list(meta, src=src_rank)
    m = meta[0]
    via_host = dist.get_backend() == "gloo"
    arrays = {}
    for n in names:
        shape, dt = m["arrays"][n]
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        where = "cpu" if via_host else device
        if rank == src_rank:
            t = torch.from_numpy(getattr(fi, n).reshape(-1).view(np.uint8)).to(where)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=where)
        dist.broadcast(t, src=src_rank)
        arrays[n] = t.cpu().numpy().view(np.dtype(dt)).reshape(shape)
        del t
    return FlatIndex(dim=m["dim"], metric=m["metric"], code_kind=m["code_kind"], threshold=m["threshold"], **arrays)


def host_cpu_info():
    """What the CPU baseline ran on: CPUs of the machine, CPUs this process may use, threads used."""
    nproc = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = nproc
    # the GPU box gives a one-GPU job a share of 16 CPUs of the machine (OMP_NUM_THREADS is set to it)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(allowed, share if share > 0 else 16))
    return nproc, allowed, threads


def cpu_baseline(fi, queries_np, k, probes, rf, budget_s=12.0):
    """The CPU oracle (a C port of the reference's search path, one query per OpenMP thread) timed
    on this host's cores over a bounded sample of the same queries."""
    from oracle import oracle as orc  # test infrastructure; used here only as the timed CPU baseline

    nproc, allowed, threads = host_cpu_info()
    oi = orc.OracleIndex(fi)
    n = min(32, queries_np.shape[0])
    t0 = time.perf_counter()
    oi.search_batch(queries_np[:n], k, probes, rf, 0, threads)
    dt = time.perf_counter() - t0
    per_q = dt / n
    sample = int(min(queries_np.shape[0], max(n, budget_s / max(per_q, 1e-9))))
    # at least ~1 s of wall time: repeat the sample if it is the whole query set
    passes, dt = 0, 0.0
    while passes == 0 or (dt < 1.0 and passes < 64):
        t0 = time.perf_counter()
        rc, ids, _, _ = oi.search_batch(queries_np[:sample], k, probes, rf, 0, threads)
        dt += time.perf_counter() - t0
        passes += 1
    # the reference's own shape: one search() after another, parallel inside (about 3 s of it)
    n_seq = int(max(8, min(sample, 3.0 / max(per_q / threads, 1e-9) / 4)))
    t0 = time.perf_counter()
    _, ids_seq, _, _ = oi.search_sequential(queries_np[:n_seq], k, probes, rf, 0, threads)
    dt_seq = time.perf_counter() - t0
    return dict(value=sample * passes / dt, unit="queries/s", cores=threads, kind="port",
                host_cpus=nproc, cpus_available_to_this_process=allowed,
                cores_note=f"{threads} OpenMP threads = this job's CPU share of the host ({nproc} CPUs in the machine, "
                           f"{allowed} in this process's affinity mask)",
                sample=f"{sample} of the {queries_np.shape[0]} queries of the first batch x {passes} passes, same index and "
                       f"parameters, {dt:.1f} s wall, oracle/cuddle_oracle.c with OpenMP (one query per thread)",
                reference_shaped=dict(value=n_seq / dt_seq, unit="queries/s", cores=threads,
                                      mean_latency_ms=dt_seq / n_seq * 1e3,
                                      identical_to_batch_parallel=bool(np.array_equal(ids_seq, ids[:n_seq])),
                                      sample=f"{n_seq} queries issued one after another, each search() spreading its routing "
                                             f"(> 100 nodes), code scan and rerank loops over the {threads} threads, as the "
                                             f"reference's rayon calls do")), ids, sample


def oracle_spot_check(dti, dev_index, queries, k, probes, rf, gpu_ids, gpu_dist, gpu_cnt, n_sample=32):
    """Full-size check of an index that exists only on the device (tree by construction): for a sample of the
    timed queries the leaves the GPU routing reaches are copied to the host - the whole centroid tree, and the ids,
    codes and fp32 rows of those leaves only; every other leaf stays in the tree with no vectors - and the CPU oracle
    (test infrastructure, used here as the checker and as a timed CPU baseline) searches that index from the
    roots. Equal ids, counts and distance bit patterns mean: same routing over the full tree (a leaf the oracle
    reaches and the GPU did not would be empty here and change the result), same selection, same rerank.
    Semantics checked: /root/reference/src/index/hierarchical.rs:330-464 (RaBitQ twin hierarchical_with_rabitq.rs:329-459)."""
    import torch

    from oracle import oracle as orc  # test infrastructure; checker + CPU baseline only
    from vectordb_amd import _ffi
    from vectordb_amd.flat_index import FlatIndex

    n = min(n_sample, queries.shape[0])
    q = queries[:n].contiguous()
    leaves, lcount = dev_index.route_device(q, probes)
    torch.cuda.synchronize()
    leaves, lcount = leaves.cpu().numpy(), lcount.cpu().numpy()
    reached = np.unique(np.concatenate([leaves[i, :lcount[i]] for i in range(n)])).astype(np.int64)
    vo = dti.vec_offsets.astype(np.int64)
    n_nodes = vo.shape[0] - 1
    sizes = np.zeros(n_nodes, dtype=np.int64)
    sizes[reached] = vo[reached + 1] - vo[reached]
    new_off = np.zeros(n_nodes + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(sizes)
    pos = np.concatenate([np.arange(vo[v], vo[v + 1]) for v in reached]) if reached.size else np.zeros(0, np.int64)
    pos_t = torch.from_numpy(pos).to(dti.vectors.device)
    take = lambda t: None if t is None else t[pos_t].cpu().numpy()  # noqa: E731
    rq = {}
    if dti.code_kind == _ffi.CODE_RABITQ:
        rq = dict(rabitq_rot_t=dti.rot_t.cpu().numpy(), rabitq_codes=take(dti.rq_codes).view(np.uint32),
                  rabitq_norm=take(dti.rq_norm), rabitq_ip_oo=take(dti.rq_ipoo))
    codes = take(dti.codes)
    part = FlatIndex(dim=dti.dim, metric=dti.metric, code_kind=dti.code_kind, centroids=dti.centroids.cpu().numpy(),
                     child_offsets=dti.child_offsets, children=dti.children, roots=dti.roots, vec_offsets=new_off,
                     ids=take(dti.ids).astype(np.uint64), vectors=take(dti.vectors), threshold=dti.threshold,
                     codes=None if codes is None else codes.view(np.uint64), **rq)
    _, _, threads = host_cpu_info()
    oi = orc.OracleIndex(part)
    qh = q.cpu().numpy()
    t0 = time.perf_counter()
    rc, ids, dist, cnt = oi.search_batch(qh, k, probes, rf, 0, threads)
    dt = time.perf_counter() - t0
    g_ids = gpu_ids[:n].cpu().numpy().astype(np.uint64)
    g_dist = np.ascontiguousarray(gpu_dist[:n].cpu().numpy(), dtype=np.float32)
    g_cnt = gpu_cnt[:n].cpu().numpy().astype(np.uint32)
    ok_cnt = bool(rc == 0 and np.array_equal(cnt, g_cnt))
    mask = np.arange(k)[None, :] < cnt[:, None]
    ok_ids = ok_cnt and bool(np.array_equal(ids[mask], g_ids[mask]))
    ok_dist = ok_cnt and bool(np.array_equal(dist.view(np.uint32)[mask], g_dist.view(np.uint32)[mask]))
    return dict(value=n / dt, unit="queries/s", cores=threads, kind="port",
                sample=f"{n} queries of the first batch, one per OpenMP thread, {dt:.2f} s wall, on a host copy of the index that "
                       f"holds the whole centroid tree ({n_nodes} nodes) and the {reached.size} leaves ({pos.size} vectors) these "
                       f"queries reach; the oracle routes from the roots like the GPU",
                gpu_ids_identical_on_sample=ok_ids, gpu_dist_bits_identical_on_sample=ok_dist,
                oracle_rc=int(rc), leaves_copied=int(reached.size), vectors_copied=int(pos.size))


def merged_ground_truth(local_rows, local_dist, local_ids, k, world):
    """Global top-k ids from every rank's local brute-force top-k (rows of its shard -> ids)."""
    import torch
    import torch.distributed as dist

    ids = local_ids[local_rows.clamp_min(0)]
    ids = torch.where(local_rows < 0, torch.full_like(ids, -1), ids)
    if world == 1:
        return ids.cpu().numpy()
    all_i = torch.empty((world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
    all_d = torch.empty((world,) + tuple(local_dist.shape), dtype=local_dist.dtype, device=ids.device)
    if dist.get_backend() == "gloo":
        hi, hd = all_i.cpu(), all_d.cpu()
        dist.all_gather(list(hi.unbind(0)), ids.cpu())
        dist.all_gather(list(hd.unbind(0)), local_dist.cpu())
        all_i, all_d = hi.to(ids.device), hd.to(ids.device)
    else:
        dist.all_gather_into_tensor(all_i.view(-1, ids.shape[1]), ids.contiguous())
        dist.all_gather_into_tensor(all_d.view(-1, ids.shape[1]), local_dist.contiguous())
    nq = ids.shape[0]
    d = all_d.permute(1, 0, 2).reshape(nq, -1)
    i = all_i.permute(1, 0, 2).reshape(nq, -1)
    _, sel = torch.topk(d, k, dim=1, largest=False)
    return torch.gather(i, 1, sel).cpu().numpy()


def run_config(name, args, rank, world, local_rank, device, backend, steps, warmup, want_cpu, want_filter):
    """Builds the configuration's index, finds the operating point, times `steps` steps and returns rank 0's
    result dictionary (None on the other ranks)."""
    import torch
    import torch.distributed as dist

    from vectordb_amd import _ffi
    from vectordb_amd.builder import build_index, metric_id
    from vectordb_amd.datasets import brute_force_topk, recall_at_k, synthetic_mixture
    from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
    from vectordb_amd.index import DeviceIndex
    from vectordb_amd.sharded import ExactShardedSearcher, ShardedSearcher

    cfg = dict(CONFIGS[name])
    if args.n and "tree" not in cfg:
        cfg["n"] = args.n
    k, probes = cfg["k"], (args.probes or cfg["probes"])
    by_construction = "tree" in cfg
    nq_config = cfg["n_queries"]
    # configs built from a data matrix (a fixed 1M-vector index): with N ranks every rank brings a batch of its own,
    # so the job's batch is N x the config's and a GPU's share of the work (routing one batch, scanning 1/N of the
    # leaves for N batches, reranking one batch's worth of rows) is what --gpus 1 measures: weak scaling.
    # --batch-scaling fixed keeps the one batch and splits it: strong scaling, where every fixed cost shows.
    per_gpu_batches = (not by_construction) and args.batch_scaling == "per-gpu"
    nq = nq_config * (world if per_gpu_batches else 1)

    # ---- data + index ----------------------------------------------------------------------------
    fi = None
    t_setup = time.perf_counter()
    dev_index = None
    if by_construction:
        # generated and indexed on the device, never copied through the host; with several ranks every rank
        # generates the leaves it owns (leaf j -> rank j mod N) and the index is N shards of cfg["n"] vectors
        from vectordb_amd.synthetic import build_tree_index_on_device
        fanout, levels, leaf_size = cfg["tree"]
        log(f"generating {cfg['label']}" + (f" x {world} ranks" if world > 1 else ""))
        dti, all_q = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf_size, device, seed=0xC0DD1E,
                                                n_queries=nq * N_BATCHES, metric=metric_id(cfg["metric"]),
                                                style=cfg.get("style", "sift"),
                                                code_kind=_ffi.CODE_RABITQ if cfg["code_kind"] == "rabitq" else _ffi.CODE_BINARY,
                                                rank=rank, world=world)
        log("ground truth (GPU brute force) on a query sample")
        gq = min(cfg.get("gt_queries", nq), nq)
        rows, dd = brute_force_topk(dti.vectors, all_q[:gq], k, cfg["metric"], chunk=1 << 20, return_dist=True)
        gt = merged_ground_truth(rows, dd, dti.ids, k, world)
        del rows, dd
        dev_index = dti.create(local_rank)
        log(f"index: {dti.centroids.shape[0]} nodes, {dti.n_vectors} vectors on this rank, threshold {dti.threshold:.6g}")
        n_total = dti.n_vectors * world
    else:
        gt = None
        if rank == 0:
            log(f"generating {cfg['label']}: n={cfg['n']} dim={cfg['dim']}")
            base, all_q = synthetic_mixture(cfg["n"], cfg["dim"], seed=0xC0DD1E, device=device, style=cfg["style"],
                                            n_queries=nq * N_BATCHES, latent_dim=cfg["latent_dim"],
                                            query_block=nq_config * N_BATCHES)
            log("ground truth (GPU brute force)")
            # recall, and with it the operating point, is judged on the config's own first batch at every N
            gt = brute_force_topk(base, all_q[:nq_config], k, cfg["metric"]).cpu().numpy()
            log("building index (GPU k-means tree)")
            # binary threshold = exact mean: the reference's single-precision running sum saturates long
            # before 10^9 values (it would give 0.017 instead of ~0.4 here and set every code bit)
            fi = build_index(base, cfg["branching"], cfg["leaf"], cfg["metric"], 20, seed=1, code_kind=cfg["code_kind"],
                             threshold=args.threshold if args.threshold != "reference" else None)
            ls = fi.leaf_sizes()
            log(f"index: {fi.n_nodes} nodes, {len(ls)} leaves (min {ls.min()}, mean {ls.mean():.1f}, max {ls.max()}), "
                f"{fi.n_levels()} levels, threshold {fi.threshold:.6g}")
            del base
            torch.cuda.empty_cache()
        else:
            all_q = torch.empty((nq * N_BATCHES, cfg["dim"]), dtype=torch.float32, device=device)
        if world > 1:
            if backend == "nccl":
                dist.broadcast(all_q, src=0)
            else:
                h = all_q.cpu()
                dist.broadcast(h, src=0)
                all_q.copy_(h)
            fi = broadcast_flat(fi, 0, device)
        dev_index = DeviceIndex(fi.shard(rank, world), local_rank)
        n_total = cfg["n"]
    batches = [all_q[b * nq:(b + 1) * nq].contiguous() for b in range(N_BATCHES)]
    log(f"setup {time.perf_counter() - t_setup:.1f} s; device bytes {dev_index.device_bytes / 1e9:.2f} GB")

    # Per-stage HIP events cost about 5 us of idle GPU per stage boundary (0.758 -> 0.732 ms per gist1m step without
    # them), so the timed loop records them on every STAGE_SAMPLE-th search only: the stage and scan-kernel durations
    # are averages over those searches of the timed region, and the other searches run as a caller's would.
    flags_box = [_ffi.SEARCH_TIME_STAGES]
    out_ids = torch.empty((nq, k), dtype=torch.int64, device=device)
    out_dist = torch.empty((nq, k), dtype=torch.float32, device=device)
    out_cnt = torch.empty((nq,), dtype=torch.int32, device=device)
    rf_box = [cfg["rerank_sweep"][0]]
    index_box = [dev_index]

    def local_search(q, kk, pp, rr):
        return index_box[0].search_batch_device(q, kk, pp, rr, flags_box[0], out_ids, out_dist, out_cnt)

    class _ShardEngine:  # this rank's shard behind the three-step interface, stage timing on
        def shard_scan(self, q, kk, pp, rr, leaves=None, leaf_count=None):
            return dev_index.shard_scan(q, kk, pp, rr, flags_box[0], leaves=leaves, leaf_count=leaf_count)

        def route_device(self, q, pp, out=None):
            return dev_index.route_device(q, pp, flags_box[0], out=out)

        def shard_scan_keys(self, q, kk, pp, rr, leaves=None, leaf_count=None):
            return dev_index.shard_scan_keys(q, kk, pp, rr, flags_box[0], leaves=leaves, leaf_count=leaf_count)

        selects_by_keys = dev_index.selects_by_keys  # RaBitQ codes: selection keys are exchanged, not histograms
        leaf_stride = staticmethod(dev_index.leaf_stride)
        packed_len = staticmethod(dev_index.packed_len)
        shard_ties = staticmethod(dev_index.shard_ties)
        shard_finish = staticmethod(dev_index.shard_finish)
        shard_finish_keys = staticmethod(dev_index.shard_finish_keys)

    exact = world > 1 and args.sharded == "exact"
    # CUDDLE_BENCH_SHARD_ROUTING=0: every rank routes the whole batch (A/B against routing a slice each + all-gather)
    clock = None
    if exact:
        from vectordb_amd.sharded import CommClock
        clock = CommClock()  # every collective of the protocol timed by name (HIP events on the stream / host clock under gloo)
    searcher = (ExactShardedSearcher(_ShardEngine(), shard_routing=os.environ.get("CUDDLE_BENCH_SHARD_ROUTING", "1") != "0",
                                     hist_exchange=args.hist_exchange, clock=clock)
                if exact else ShardedSearcher(local_search))
    if exact and args.collectives == "cabi":
        # the same protocol with RCCL called by the library (cuddle_comm_* / cuddle_sharded_search_batch); the id
        # made on rank 0 travels over the process group that is already up
        from vectordb_amd.sharded import RcclShardedSearcher
        if backend != "nccl":
            raise SystemExit("--collectives cabi needs one GPU per rank (RCCL cannot place two ranks on one device)")
        box = [RcclShardedSearcher.make_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rccl = RcclShardedSearcher(dev_index, rank, world, box[0])

        class _CabiSearcher:
            def search_batch(self, q, kk, pp, rr):
                return rccl.search_batch(q, kk, pp, rr, flags_box[0], out_ids, out_dist, out_cnt)

        searcher = _CabiSearcher()
    step_no = [0]

    def step(batch=None):
        b = step_no[0] % N_BATCHES if batch is None else batch
        flags_box[0] = _ffi.SEARCH_TIME_STAGES if step_no[0] % STAGE_SAMPLE == 0 else 0
        step_no[0] += 1
        out = searcher.search_batch(batches[b], k, probes, rf_box[0])
        if world > 1 and backend != "nccl":
            torch.cuda.synchronize()  # rehearsal backends stage CUDA tensors through the host on their own streams
        return out

    def recall_of(ids):
        return recall_at_k(ids.cpu().numpy()[:len(gt)], gt, k)

    # ---- operating point: cheapest sweep point with recall@k >= 0.95 (first batch) ------------------
    sweep = []
    chosen = None
    for rf in ([args.rerank] if args.rerank else cfg["rerank_sweep"]):
        rf_box[0] = rf
        ids, _, _ = step(0)
        torch.cuda.synchronize()
        if rank == 0:
            r = recall_of(ids)
            sweep.append((rf, r))
            log(f"probes={probes} rerank_factor={rf}: recall@{k} = {r:.4f}")
        flag = torch.tensor([1 if (rank == 0 and sweep[-1][1] >= 0.95) else 0])
        if world > 1:
            if backend == "nccl":
                flag = flag.to(device)
            dist.broadcast(flag, src=0)
        if int(flag.item()):
            chosen = rf
            break
    if chosen is None:
        chosen = (args.rerank or cfg["rerank_sweep"][-1])
    rf_box[0] = chosen
    trace(f"operating point rf={chosen}")
    dev_index.stage_times()  # drop the sweep's timings

    # ---- timed region ---------------------------------------------------------------------------
    for _ in range(warmup):
        step()
    dev_index.stage_times()
    if clock is not None:
        clock.reset()
    step_no[0] = 0  # the timed region starts on a sampled search
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64)
        if backend == "nccl":
            te = te.to(device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    stage = dev_index.stage_times()
    stats = dev_index.last_stats()
    comm = None
    if clock is not None and args.collectives == "torch":
        # this rank's collectives over the timed region: milliseconds per call, bytes this rank contributes per call
        comm = {"per_collective": clock.summary(), "hist_exchange": args.hist_exchange,
                "note": "rank 0's view; ms_per_call includes waiting for the slowest rank to arrive"}
        comm["ms_per_step"] = sum(v["ms_per_call"] * v["calls"] for v in comm["per_collective"].values()) / max(steps, 1)
        clock.reset()
        clock = searcher.clock = None  # the loops below run untimed

    # per-batch latency distribution (separate loop: one sync per batch), at least 200 samples
    lat = []
    for _ in range(max(steps, 200)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()
    dev_index.stage_times()

    # N > 1 with a batch per GPU: the one-batch (strong-scaling) case beside it, outside the timed region
    one_batch = None
    if world > 1 and per_gpu_batches:
        small = [b[:nq_config].contiguous() for b in batches]
        for i in range(3):
            searcher.search_batch(small[i % N_BATCHES], k, probes, chosen)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            searcher.search_batch(small[i % N_BATCHES], k, probes, chosen)
            if backend != "nccl":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        dt = (time.perf_counter() - t1) / steps
        one_batch = {"queries": nq_config, "ms_per_batch": dt * 1e3, "queries_per_s": nq_config / dt,
                     "note": f"the config's single {nq_config}-query batch split over the {world} ranks (strong scaling), this rank's clock"}
        dev_index.stage_times()

    ids, dd, cc = step(0)  # every rank: the search is collective
    torch.cuda.synchronize()
    if rank != 0:
        return None
    recall = recall_of(ids)
    sample_out = (ids[:64].clone(), dd[:64].clone(), cc[:64].clone())  # later searches reuse the output buffers

    # the host-pointer entry point (queries and results cross PCIe inside the call): reported beside `value`, never as it
    host_entry = None
    if world == 1:
        qh = batches[0].cpu().numpy()
        dev_index.search_batch(qh, k, probes, chosen, 0)
        t1 = time.perf_counter()
        for _ in range(10):
            dev_index.search_batch(qh, k, probes, chosen, 0)
        dt = (time.perf_counter() - t1) / 10
        host_entry = {"queries_per_s": nq / dt, "ms_per_batch": dt * 1e3,
                      "note": "cuddle_search_batch on host buffers: pageable H2D of the queries, the search, D2H of ids/distances/counts, synchronous"}
        dev_index.stage_times()

    # ---- roofline of the dominant kernel (the code scan + select) --------------------------------
    rabitq = cfg["code_kind"] == "rabitq"
    code_bytes = (4 * ((cfg["dim"] + 31) // 32) + 8) if rabitq else 8 * ((cfg["dim"] + 63) // 64)
    scan_kernel = "scan_select_rabitq_kernel" if rabitq else "scan_select_hamming_kernel"
    # binary codes of 512 bits and more, batches that reach a leaf several times over: the scan goes leaf by leaf
    # (leaf_hamming_kernel reads a leaf's codes once per 64 queries, select_rows_kernel selects from the distances it
    # hands over; item_scatter_kernel, where it runs, orders the work items); the scan stage is then those launches
    scan_mode = dev_index.scan_mode(nq, probes) if not rabitq else "per_query"
    if scan_mode == "leaf_by_leaf":
        scan_kernel = "leaf_hamming_kernel + select_rows_kernel"
    scan_bytes = stats["vectors_scanned"] * code_bytes           # algorithmic bytes of one launch (this rank, last step)
    rerank_bytes = stats["vectors_reranked"] * 4 * cfg["dim"]
    n_timed = max(stage["searches"], 1)
    scan_ms = stage["scan"] / n_timed
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_hbm_traffic_{name}.json")
    if world == 1 and os.path.exists(pmc_path):  # measured separately under rocprofv3 --pmc (it cannot run inside this process)
        pmc = json.load(open(pmc_path))
        parts = [pmc["kernels"].get(kn.strip(), {}).get("traffic_bytes_per_launch") for kn in scan_kernel.split("+")]
        traffic = sum(parts) if all(p is not None for p in parts) else None
        if traffic is not None and scan_mode == "leaf_by_leaf":  # (runs only where items differ enough in size to be ordered)
            traffic += pmc["kernels"].get("item_scatter_kernel", {}).get("traffic_bytes_per_launch") or 0.0
        traffic_src = os.path.relpath(pmc_path, ROOT)
    code_array = dev_index_codes_bytes = (n_total // world) * code_bytes
    fits_l3 = code_array <= (256 << 20)
    roofline = dict(bound="hbm", kernel=scan_kernel, scan_mode=scan_mode, achieved=achieved, peak=HBM_PEAK_GBPS,
                    unit="GB/s", frac=achieved / HBM_PEAK_GBPS, traffic=traffic, traffic_source=traffic_src,
                    traffic_kind="L2-miss (fabric) bytes per launch from TCC FETCH_SIZE + WRITE_SIZE; FETCH_SIZE also counts "
                                 "requests served by the 256 MiB Infinity Cache, so this is HBM traffic only for arrays "
                                 "that cannot sit in that cache",
                    algorithmic_bytes_per_launch=scan_bytes, avg_launch_ms=scan_ms,
                    stage_ms_per_step={n: stage[n] / n_timed for n in _ffi.STAGE_NAMES},
                    rerank_GBps=(rerank_bytes / (stage["rerank"] / n_timed * 1e-3) / 1e9) if stage["rerank"] > 0 else 0.0,
                    code_array_bytes=code_array, code_array_fits_infinity_cache=fits_l3,
                    note=("leaf by leaf: achieved counts the reference's bytes (every query reads the codes of its own candidates, "
                          "hierarchical.rs:591-641) over the time of the two launches that replace those reads - a leaf's codes are "
                          "fetched once per 64 queries that reach it, so the figure may pass what memory could deliver per query "
                          "(frac > 1) and `traffic` is far below the algorithmic bytes; what limits the launches is integer issue "
                          "(one xor and one counting add per 32 code bits and pair); the selection reads the distances back from the spill rows; "
                          if scan_mode == "leaf_by_leaf" else "") +
                         ("the whole code array fits the 256 MiB Infinity Cache: achieved is on-die cache bandwidth as much as "
                          "HBM bandwidth (see roofline_hbm_resident for the same kernel on a code array that does not fit); "
                          if fits_l3 else "the code array is larger than the 256 MiB Infinity Cache: the scan streams from HBM; ") +
                         f"achieved = algorithmic code bytes / mean duration of the scan launch(es) alone (HIP events around them, on every {STAGE_SAMPLE}th search of the timed loop, "
                         f"which runs its stages on one stream; the candidate-count / block-order / pair-order kernels before them are stage scan_prep); "
                         f"the timed loop rotates {N_BATCHES} query batches")
    del dev_index_codes_bytes
    # what the scan's reads alone cost: a kernel that issues exactly the scan's loads (same leaves, same pieces, as many
    # in flight) and nothing else, timed back to back with the scan kernel on the first batch
    if world == 1 and not rabitq:
        try:
            nqc = min(nq, 4096)  # one chunk per batch
            rot = torch.stack([b[:nqc] for b in batches]).contiguous()
            read_ms, scan_ms_same, code_bytes = dev_index.scan_read_ceiling(rot, k, probes, chosen, 5)
            del rot
            roofline.update(ceiling_GBps=code_bytes / (read_ms * 1e-3) / 1e9, ceiling_launch_ms=read_ms,
                            scan_launch_ms_same_run=scan_ms_same, frac_of_ceiling=read_ms / scan_ms_same,
                            per_query_scan_GBps=code_bytes / (scan_ms_same * 1e-3) / 1e9,
                            per_query_scan_frac=code_bytes / (scan_ms_same * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            ceiling_queries_per_launch=nqc,
                            ceiling_note="per_query_scan_* = scan_select_hamming_kernel (one block per query, every query reading its own candidates' codes: the kernel the "
                                         "algorithmic-bytes figure is defined on) timed here on the same batches, whichever scan the timed loop uses; "
                                         "ceiling = the per-query scan's own reads with no arithmetic, histogram or selection behind them "
                                         "(scan_read_ceiling_kernel: same grid, block, LDS size, tables and loads); both kernels "
                                         f"timed with HIP events over 5 rounds of the same {N_BATCHES} rotating batches")
        except _ffi.CuddleError as e:
            roofline.update(ceiling_GBps=None, ceiling_note=str(e))
        dev_index.stage_times()

    # routing filter: approximate (matrix-core) distances against the error bound its proof uses
    route_filter = []
    if world == 1 and want_filter:
        q_host = batches[0][:256].cpu().numpy()
        for level in range(dev_index.levels):
            try:
                ratio, err = dev_index.route_filter_error(q_host, probes, level)
                route_filter.append({"level": level, "max_err_over_bound": ratio, "max_abs_err": err})
            except _ffi.CuddleError:
                pass  # narrow level (exact kernels only) or never a whole frontier

    cpu = None
    if want_cpu and fi is not None:
        log("CPU baseline (oracle port, OpenMP)")
        cpu, cpu_ids, sample = cpu_baseline(fi, batches[0].cpu().numpy(), k, probes, chosen)
        same = bool(np.array_equal(cpu_ids[:sample].astype(np.int64), ids.cpu().numpy()[:sample])) if (world == 1 or exact) else None
        cpu["gpu_ids_identical_on_sample"] = same
    elif want_cpu and by_construction and world == 1:
        log("full-size oracle spot check (host copy of the reached leaves)")
        cpu = oracle_spot_check(dti, dev_index, batches[0], k, probes, chosen, *sample_out)

    # the reference's own threshold (binary.rs:122-141: a sequential f32 sum) beside the exact mean, same operating point
    ref_threshold = None
    if (world == 1 and fi is not None and cfg["code_kind"] == "binary" and args.threshold == "mean"
            and not args.no_reference_threshold):
        thr_ref = float(_ffi.lib().cuddle_binary_threshold(_ffi.ptr(np.ascontiguousarray(fi.vectors)), fi.n_vectors, fi.dim))
        alt = FlatIndex(dim=fi.dim, metric=fi.metric, code_kind=fi.code_kind, centroids=fi.centroids,
                        child_offsets=fi.child_offsets, children=fi.children, roots=fi.roots, vec_offsets=fi.vec_offsets,
                        ids=fi.ids, vectors=fi.vectors, threshold=thr_ref, codes=pack_binary_codes(fi.vectors, thr_ref))
        index_box[0] = DeviceIndex(alt, local_rank)
        ids_ref, _, _ = step(0)
        torch.cuda.synchronize()
        ref_threshold = dict(threshold=thr_ref, exact_mean=fi.threshold, rerank_factor=chosen,
                             **{f"recall_at_{k}": recall_of(ids_ref)})
        index_box[0].close()
        index_box[0] = dev_index

    total_q = nq * steps
    out = {
        "metric": f"batched QPS at recall@{k}>=0.95",
        "value": total_q / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (by_construction or per_gpu_batches) else "strong",
        "vs_baseline": None,
        "dtype": "f32 sign-masked sums (RaBitQ estimator) + f32 rerank" if rabitq else "u64 xor/popcount scan + f32 rerank",
        "data": "synthetic",
        "config": {"workload": cfg["label"] + f", {cfg['metric']}, k={k}, {nq}-query batch" +
                               (f" ({nq_config} per GPU)" if world > 1 and per_gpu_batches else "") + f", probes_per_level={probes}, "
                               f"rerank_factor={chosen}" + (f", branching {cfg['branching']} / leaf {cfg['leaf']}" if "branching" in cfg else ""),
                   "name": name, "binary_threshold": args.threshold, "index_vectors": n_total,
                   "parallelism": (f"leaf-sharded x{world}, {args.sharded} selection" if world > 1 else "single GPU")},
        f"recall_at_{k}": recall,
        "recall_sweep": sweep,
        "host_pointer_entry": host_entry,
        "one_batch_split": one_batch,
        "comm_ms": comm,
        "reference_threshold": ref_threshold,
        "p50_batch_ms": lat[len(lat) // 2],
        "p99_batch_ms": lat[min(len(lat) - 1, len(lat) * 99 // 100)],
        "latency_samples": len(lat),
        "candidates_per_query": stats["vectors_scanned"] / nq,
        "leaves_per_query": stats["leaves_searched"] / nq,
        "reranked_per_query": stats["vectors_reranked"] / nq,
        "route_fallbacks_per_batch": stats["route_fallbacks"],
        "route_filter": route_filter,
        "max_candidates": stats["max_candidates"],
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    dev_index.close()
    return out


def compact_config(big, wall_s):
    """One configuration's result, cut down to what the `configs` block of the headline line carries."""
    k = [key for key in big if key.startswith("recall_at_")][0]
    r = big["roofline"]
    cpu = big.get("cpu_baseline")
    if cpu:
        cpu = {key: cpu[key] for key in ("value", "unit", "cores", "kind", "sample", "gpu_ids_identical_on_sample",
                                         "gpu_dist_bits_identical_on_sample") if key in cpu}
    return {"workload": big["config"]["workload"], "queries_per_s": big["value"], "ms_per_step": big["ms_per_step"],
            "steps": big["steps"], "warmup": big["warmup"], k: big[k], "recall_sweep": big["recall_sweep"],
            "p50_batch_ms": big["p50_batch_ms"], "p99_batch_ms": big["p99_batch_ms"],
            "candidates_per_query": big["candidates_per_query"], "reranked_per_query": big["reranked_per_query"],
            "route_fallbacks_per_batch": big["route_fallbacks_per_batch"], "dtype": big["dtype"],
            "stage_ms": r["stage_ms_per_step"],
            "roofline": {key: r[key] for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                 "algorithmic_bytes_per_launch", "avg_launch_ms", "rerank_GBps",
                                                 "code_array_bytes", "code_array_fits_infinity_cache", "ceiling_GBps",
                                                 "frac_of_ceiling", "scan_mode", "per_query_scan_GBps", "per_query_scan_frac") if key in r},
            "cpu_baseline": cpu, "wall_s": wall_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="gist1m", choices=sorted(CONFIGS))
    ap.add_argument("--n", type=int, default=0, help="override the number of base vectors (smoke runs)")
    ap.add_argument("--rerank", type=int, default=0, help="fix rerank_factor instead of sweeping")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true",
                    help="skip the extra measurement of the scan on the 10M x 960 index (default workload, one GPU)")
    ap.add_argument("--no-reference-threshold", action="store_true")
    ap.add_argument("--threshold", default="mean", choices=["mean", "reference"],
                    help="binary threshold: exact mean, or the reference's saturating f32 running sum")
    ap.add_argument("--probes", type=int, default=0)
    ap.add_argument("--batch-scaling", default="per-gpu", choices=["per-gpu", "fixed"],
                    help="N > 1, configs built from a data matrix: per-gpu = the job's batch is N x the config's (weak scaling, "
                         "default); fixed = the config's one batch split over the ranks (strong scaling)")
    ap.add_argument("--collectives", default="torch", choices=["torch", "cabi"],
                    help="N > 1, exact: who calls the collectives - torch.distributed (default) or the library itself through cuddle_comm_*")
    ap.add_argument("--hist-exchange", default="auto", choices=["auto", "full", "coarse_fine"],
                    help="N > 1, exact, binary codes: how the Hamming histograms travel - whole ([Q][dim + 1] u32), or in bins of 16 "
                         "values and then the 16 values of the threshold's bin (two small all-reduces); auto = the latter from 256 bins on")
    ap.add_argument("--sharded", default="exact", choices=["exact", "simple"],
                    help="N > 1: 'exact' reproduces the single-index result bit for bit (two small all-reduces + one "
                         "all-gather per batch); 'simple' lets every rank rerank its own top candidates (one all-gather)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # this process stays off the GPU; the ranks are fresh interpreters

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    # one GPU per rank; CUDDLE_BENCH_BACKEND=gloo lets several ranks share a GPU to rehearse the N > 1 flow on a 1-GPU box
    backend = os.environ.get("CUDDLE_BENCH_BACKEND", "nccl")
    local_rank = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    t_start = time.perf_counter()
    out = run_config(args.config, args, rank, world, local_rank, device, backend, args.steps, args.warmup,
                     want_cpu=not args.no_cpu_baseline, want_filter=True)
    if rank == 0 and world == 1 and args.config == "gist1m" and not args.no_hbm_probe and not args.n:
        # The other BASELINE.json configurations in the same run, each through the same run_config (same timing
        # contract, same operating-point rule), summarised under "configs". `value` stays config #2 (the configuration
        # the metric is quoted on: GIST1M at the reference's precision on one GPU); #3 rabitq10m, #1 sift1m (the
        # reference's CPU-runnable case, with its own CPU baseline) and one GPU's shard of #4 / #5 ride along.
        # Budget: a configuration is started only while the run is inside CUDDLE_BENCH_BUDGET_S (default 400 s).
        budget = float(os.environ.get("CUDDLE_BENCH_BUDGET_S", "400"))
        configs, skipped = {}, []
        for name in ["gist10m", "sift1m", "rabitq10m", "shard12m", "shard125m"]:
            torch.cuda.empty_cache()
            if time.perf_counter() - t_start > budget:
                skipped.append(name)
                continue
            log(f"config {name}")
            t_cfg = time.perf_counter()
            try:
                big = run_config(name, args, rank, world, local_rank, device, backend, args.steps, args.warmup,
                                 want_cpu=not args.no_cpu_baseline, want_filter=False)
            except Exception as e:  # a configuration that cannot run (memory, a failed launch) must not cost the headline line
                log(f"config {name} failed: {type(e).__name__}: {e}")
                skipped.append(f"{name} ({type(e).__name__}: {e})"[:200])
                continue
            configs[name] = compact_config(big, time.perf_counter() - t_cfg)
            if name == "gist10m":
                # the same kernels where the codes cannot sit in the Infinity Cache: 10M x 960, 1.2 GB of codes
                r = dict(big["roofline"])
                # the roofline's own "unit" (GB/s) stays; the probe's throughput goes beside it under its own names
                r.update(workload=big["config"]["workload"], queries_per_s=big["value"], ms_per_step=big["ms_per_step"],
                         recall_at_10=big["recall_at_10"], candidates_per_query=big["candidates_per_query"])
                out["roofline_hbm_resident"] = r
        out["configs"] = configs
        out["configs_skipped"] = skipped
        out["bench_wall_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
