"""Per-rank generation of a leaf-sharded index (vectordb_amd.synthetic.build_tree_index_on_device with
world > 1) on CPU with two gloo ranks: every rank generates only its own leaves, the ranks exchange the
leaf centroids (all-gather) and the value sum behind the binary threshold (all-reduce), and must end up
with the same tree. The shards are then searched with the exact sharded protocol (oracle-based
engines, as in test_sharded_exact_gloo.py) and the result must equal the oracle's search of the union
of the shards, bit for bit."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharded_exact_gloo import OracleShardEngine, _free_port
from vectordb_amd.flat_index import FlatIndex

DIM, FANOUT, LEVELS, LEAF = 24, 5, 2, 9
CASES = ((10, 4, 3), (5, 50, 10), (10, 8, 1000))


def _union(shards):
    """The whole index from its leaf shards: leaf j's vectors come from shard j % world."""
    world = len(shards)
    s0 = shards[0]
    leaf = s0.is_leaf()
    leaf_no = np.cumsum(leaf) - 1
    parts = {k: [] for k in ("ids", "vectors", "codes")}
    sizes = np.zeros(s0.n_nodes, dtype=np.int64)
    for node in np.nonzero(leaf)[0]:
        sh = shards[leaf_no[node] % world]
        a, b = int(sh.vec_offsets[node]), int(sh.vec_offsets[node + 1])
        sizes[node] = b - a
        for k in parts:
            parts[k].append(getattr(sh, k)[a:b])
    off = np.zeros(s0.n_nodes + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    return FlatIndex(dim=s0.dim, metric=s0.metric, code_kind=s0.code_kind, centroids=s0.centroids,
                     child_offsets=s0.child_offsets, children=s0.children, roots=s0.roots, vec_offsets=off,
                     ids=np.concatenate(parts["ids"]), vectors=np.concatenate(parts["vectors"]), threshold=s0.threshold,
                     codes=np.concatenate(parts["codes"])).normalised()


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vectordb_amd.sharded import ExactShardedSearcher
        from vectordb_amd.synthetic import build_tree_index_on_device

        dti, q = build_tree_index_on_device(DIM, FANOUT, LEVELS, LEAF, "cpu", seed=4, n_queries=20, style="sift",
                                            rank=rank, world=world)
        fi = dti.to_flat()
        assert fi.n_vectors == FANOUT ** LEVELS * LEAF  # one rank's share of world * fanout**levels leaves
        everyone = [None] * world
        dist.all_gather_object(everyone, fi)
        for other in everyone:  # same tree, same threshold, same queries everywhere
            assert np.array_equal(other.centroids, fi.centroids) and other.threshold == fi.threshold
            assert np.array_equal(other.child_offsets, fi.child_offsets) and np.array_equal(other.roots, fi.roots)
        qs = [None] * world
        dist.all_gather_object(qs, q.numpy())
        assert all(np.array_equal(v, qs[0]) for v in qs)
        searcher = ExactShardedSearcher(OracleShardEngine(fi))
        res = [tuple(t.numpy().copy() for t in searcher.search_batch(q, k, probes, rf)) for (k, probes, rf) in CASES]
        if rank == 0:
            from oracle import oracle as orc
            full = _union(everyone)
            assert full.n_vectors == world * fi.n_vectors and len(set(full.ids.tolist())) == full.n_vectors
            assert abs(float(full.vectors.astype(np.float64).mean()) - full.threshold) < 1e-4
            oi = orc.OracleIndex(full)
            for (k, probes, rf), (ids, dd, cnt) in zip(CASES, res):
                rc, oids, odd, ocnt = oi.search_batch(q.numpy(), k, probes, rf)
                assert rc == 0 and np.array_equal(cnt, ocnt.astype(np.int32))
                assert np.array_equal(ids.view(np.uint64), oids) and np.array_equal(dd.view(np.uint32), odd.view(np.uint32))
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_generate_and_search_their_own_leaves(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
