"""The exact N > 1 protocol on CPU: two and three gloo ranks, each holding one leaf shard, run
`ExactShardedSearcher` (two all-reduces + one all-gather + ordered merge, vectordb_amd/sharded.py)
and must reproduce the unsharded oracle search bit for bit. The per-rank engine is built from the
CPU oracle's stage outputs (the GPU kernels cannot run in this container); the collectives, the
threshold / tie-budget rule as the protocol states it, and the merge are the code under test."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import queries_like, small_index


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShardEngine:
    """One shard, phases as in include/cuddle_hip.h (cuddle_shard_scan / _ties / _finish), computed
    with the oracle's candidate lists and distances."""

    def __init__(self, flat_shard):
        from oracle import oracle as orc
        self.orc = orc
        self.fi = flat_shard
        self.oi = orc.OracleIndex(flat_shard)

    def route_device(self, queries, probes):
        """Routing of a slice of the batch: (leaves [n, stride], count [n]); the stride is a property of the tree."""
        q = queries.numpy()
        self.routed = q.shape[0]
        leaves = np.zeros((q.shape[0], self.fi.n_nodes), dtype=np.int32)
        count = np.zeros(q.shape[0], dtype=np.int32)
        for i in range(q.shape[0]):
            lv = self.oi.search(q[i], 1, probes, 1, trace=True)["leaves"]
            leaves[i, :len(lv)] = lv
            count[i] = len(lv)
        return torch.from_numpy(leaves), torch.from_numpy(count)

    def shard_scan(self, queries, k, probes, rf, leaves=None, leaf_count=None):
        q = queries.numpy()
        self.k, self.rerank_k = k, min(k * rf, 10000)
        D = self.fi.dim
        self.tr = [self.oi.search(q[i], k, probes, rf, trace=True) for i in range(q.shape[0])]
        if leaves is not None:  # lists routed by the other ranks and all-gathered: they must be this rank's own routing
            self.given = True
            for i, t in enumerate(self.tr):
                assert int(leaf_count[i]) == len(t["leaves"]), i
                assert np.array_equal(leaves[i, :len(t["leaves"])].numpy().astype(np.uint32), t["leaves"]), i
        self.stride = max(len(t["leaves"]) for t in self.tr)
        hist = np.zeros((q.shape[0], D + 1), dtype=np.int32)
        for i, t in enumerate(self.tr):
            hist[i] = np.bincount(t["cand_key"], minlength=D + 1)
        return torch.from_numpy(hist)

    def _slots(self, t):
        sizes = (self.fi.vec_offsets[t["leaves"] + 1] - self.fi.vec_offsets[t["leaves"]]).astype(np.int64)
        slot = np.repeat(np.arange(len(sizes)), sizes)
        inleaf = np.concatenate([np.arange(s) for s in sizes]) if len(sizes) else np.zeros(0, dtype=np.int64)
        return slot, inleaf

    def shard_ties(self, ghist):
        g = ghist.numpy().astype(np.int64)
        nq, D1 = g.shape
        self.T, self.budget = np.zeros(nq, dtype=np.int64), np.zeros(nq, dtype=np.int64)
        ties = np.zeros((nq, self.stride), dtype=np.int32)
        for i, t in enumerate(self.tr):
            cum = np.cumsum(g[i])
            hit = np.nonzero(cum >= self.rerank_k)[0]
            if len(hit):
                self.T[i] = hit[0]
                self.budget[i] = self.rerank_k - (cum[hit[0]] - g[i][hit[0]])
            else:
                self.T[i], self.budget[i] = D1, 0
            slot, _ = self._slots(t)
            m = t["cand_key"] == self.T[i]
            ties[i, :len(t["leaves"])] = np.bincount(slot[m], minlength=len(t["leaves"]))[:len(t["leaves"])]
        return torch.from_numpy(ties)

    def shard_finish(self, queries, gties):
        q, gt = queries.numpy(), gties.numpy().astype(np.int64)
        nq, k = q.shape[0], self.k
        ids = np.full((nq, k), -1, dtype=np.int64)
        dd = np.full((nq, k), np.inf, dtype=np.float32)
        order = np.full((nq, k), -1, dtype=np.int64)
        cnt = np.zeros(nq, dtype=np.int32)
        for i, t in enumerate(self.tr):
            slot, inleaf = self._slots(t)
            key = t["cand_key"].astype(np.int64)
            before = np.concatenate([[0], np.cumsum(gt[i])])[:-1]
            quota = np.clip(self.budget[i] - before, 0, gt[i])
            tie = key == self.T[i]
            tie_rank = np.zeros(len(key), dtype=np.int64)
            for s in np.unique(slot[tie]):
                m = tie & (slot == s)
                tie_rank[m] = np.arange(m.sum())
            sel = (key < self.T[i]) | (tie & (tie_rank < quota[slot] if len(slot) else False))
            pos = t["cand_pos"][sel].astype(np.int64)
            if len(pos) == 0:
                continue
            vec = self.fi.vectors[pos]
            d = self.orc.distance_pairs(np.repeat(q[i][None], len(pos), 0), vec, int(self.fi.metric))
            o = (key[sel] << 48) | (slot[sel] << 24) | inleaf[sel]
            idx = np.lexsort((o, d))[:k]
            n = len(idx)
            ids[i, :n], dd[i, :n], order[i, :n], cnt[i] = self.fi.ids[pos[idx]].astype(np.int64), d[idx], o[idx], n
        return torch.from_numpy(ids), torch.from_numpy(dd), torch.from_numpy(order), torch.from_numpy(cnt)


    # ---- RaBitQ codes: the key protocol (cuddle_shard_scan_rabitq / cuddle_shard_finish_rabitq) ----
    @property
    def selects_by_keys(self):
        from vectordb_amd import _ffi
        return int(self.fi.code_kind) == _ffi.CODE_RABITQ

    @staticmethod
    def _f32_key(bits):
        bits = bits.astype(np.uint64)
        return np.where(bits & 0x80000000, bits ^ 0xFFFFFFFF, bits | 0x80000000)

    def shard_scan_keys(self, queries, k, probes, rf, leaves=None, leaf_count=None):
        q = queries.numpy()
        self.k, self.rerank_k = k, min(k * rf, 10000)
        self.tr = [self.oi.search(q[i], k, probes, rf, trace=True) for i in range(q.shape[0])]
        if leaves is not None:
            self.given = True
            for i, t in enumerate(self.tr):
                assert int(leaf_count[i]) == len(t["leaves"]), i
        keys = np.full((q.shape[0], self.rerank_k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        self.local = []
        for i, t in enumerate(self.tr):
            slot, inleaf = self._slots(t)
            kk = (self._f32_key(t["cand_key"]) << np.uint64(32)) | (slot.astype(np.uint64) << np.uint64(18)) | inleaf.astype(np.uint64)
            o = np.argsort(kk, kind="stable")[:self.rerank_k]
            keys[i, :len(o)] = kk[o]
            self.local.append((kk[o], t["cand_pos"][o].astype(np.int64)))
        return torch.from_numpy(keys.view(np.int64))

    def shard_finish_keys(self, queries, all_keys):
        q = queries.numpy()
        ak = all_keys.numpy().view(np.uint64)  # [world, nq, R]
        nq, k = q.shape[0], self.k
        ids = np.full((nq, k), -1, dtype=np.int64)
        dd = np.full((nq, k), np.inf, dtype=np.float32)
        order = np.full((nq, k), -1, dtype=np.int64)
        cnt = np.zeros(nq, dtype=np.int32)
        for i in range(nq):
            allk = np.sort(ak[:, i, :].reshape(-1))
            valid = allk[allk != np.uint64(0xFFFFFFFFFFFFFFFF)]
            thr = valid[self.rerank_k - 1] if len(valid) > self.rerank_k else np.uint64(0xFFFFFFFFFFFFFFFE)
            kk, pos = self.local[i]
            m = kk <= thr
            kk, pos = kk[m], pos[m]
            if len(pos) == 0:
                continue
            d = self.orc.distance_pairs(np.repeat(q[i][None], len(pos), 0), self.fi.vectors[pos], int(self.fi.metric))
            idx = np.lexsort((kk, d))[:k]
            n = len(idx)
            ids[i, :n], dd[i, :n], order[i, :n], cnt[i] = self.fi.ids[pos[idx]].astype(np.int64), d[idx], kk[idx].view(np.int64), n
        return torch.from_numpy(ids), torch.from_numpy(dd), torch.from_numpy(order), torch.from_numpy(cnt)


CASES = {"tight": (10, 4, 3), "mid": (10, 6, 10), "one": (1, 2, 1), "covering": (10, 4, 1000)}


def _worker(rank, world, port, out_path, code_kind="binary", d=16):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vectordb_amd.sharded import CommClock, ExactShardedSearcher

        x, fi = small_index(n=4000, d=d, n_clusters=20, branching=6, leaf=30, code_kind=code_kind)
        q = queries_like(x, 24, 9)
        engine = OracleShardEngine(fi.shard(rank, world))
        searcher = ExactShardedSearcher(engine, hist_exchange="full")
        assert searcher.world == world and searcher.rank == rank and searcher.shard_routing
        plain = ExactShardedSearcher(OracleShardEngine(fi.shard(rank, world)), shard_routing=False)
        # the two-step histogram exchange (bins of 16, then the 16 values of the threshold's bin), with every collective timed
        clock = CommClock()
        coarse = ExactShardedSearcher(OracleShardEngine(fi.shard(rank, world)), hist_exchange="coarse_fine", clock=clock)
        res = {}
        for name, (k, probes, rf) in CASES.items():
            ids, dd, cnt = searcher.search_batch(torch.from_numpy(q), k, probes, rf)
            per = (q.shape[0] + world - 1) // world
            assert engine.given and engine.routed == max(0, min(per, q.shape[0] - rank * per))  # routed only its slice
            ids2, dd2, cnt2 = plain.search_batch(torch.from_numpy(q), k, probes, rf)  # every rank routing everything
            assert torch.equal(ids, ids2) and torch.equal(dd, dd2) and torch.equal(cnt, cnt2)
            ids3, dd3, cnt3 = coarse.search_batch(torch.from_numpy(q), k, probes, rf)
            assert torch.equal(ids, ids3) and torch.equal(dd, dd3) and torch.equal(cnt, cnt3), name
            res[name] = (ids.numpy().copy(), dd.numpy().copy(), cnt.numpy().copy())
            t = torch.from_numpy(res[name][0].copy())
            dist.broadcast(t, src=0)
            assert torch.equal(t, torch.from_numpy(res[name][0])), f"rank {rank} differs from rank 0 ({name})"
        timed = clock.summary()
        if code_kind == "binary":
            assert set(timed) == {"leaf_gather", "hist_coarse_allreduce", "hist_fine_allreduce", "tie_allreduce", "result_gather"}, timed
            assert timed["hist_coarse_allreduce"]["bytes_per_call"] == q.shape[0] * ((d + 1 + 15) // 16) * 4
            assert timed["hist_fine_allreduce"]["bytes_per_call"] == q.shape[0] * 16 * 4
        else:
            assert set(timed) == {"leaf_gather", "key_gather", "result_gather"}, timed
        assert all(v["calls"] == len(CASES) and v["ms_per_call"] >= 0 for v in timed.values())
        if rank == 0:
            np.savez(out_path, q=q, **{f"{n}_{f}": v for n, r in res.items() for f, v in zip(("ids", "dist", "cnt"), r)})
    finally:
        dist.destroy_process_group()


def _run(world, tmp_path, code_kind="binary", d=16):
    from oracle import oracle as orc

    out = str(tmp_path / f"exact{world}.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, code_kind, d), nprocs=world, join=True)
    z = np.load(out)
    x, fi = small_index(n=4000, d=d, n_clusters=20, branching=6, leaf=30, code_kind=code_kind)
    full = orc.OracleIndex(fi)
    for name, (k, probes, rf) in CASES.items():
        rc, ids, dd, cnt = full.search_batch(z["q"], k, probes, rf)
        assert rc == 0
        assert np.array_equal(z[f"{name}_cnt"], cnt.astype(np.int32)), name
        assert np.array_equal(z[f"{name}_ids"].view(np.uint64), ids), name
        assert np.array_equal(z[f"{name}_dist"].view(np.uint32), dd.view(np.uint32)), name


def test_two_rank_exact_sharded_search(tmp_path):
    _run(2, tmp_path)


def test_three_rank_exact_sharded_search(tmp_path):
    _run(3, tmp_path)


def test_two_rank_exact_sharded_search_rabitq(tmp_path):
    """RaBitQ codes: one all-gather of selection keys instead of the two all-reduces."""
    _run(2, tmp_path, "rabitq")


def test_two_rank_coarse_then_fine_histograms_wider_codes(tmp_path):
    """71 Hamming values = 5 coarse bins: the threshold's bin lies in the middle, candidates below it are lumped."""
    _run(2, tmp_path, d=70)
