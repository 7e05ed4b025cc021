"""Leaf-sharded search over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm).

Partitioning: every rank holds the whole centroid tree (it is small) and the codes + fp32 vectors
of the leaves j with j % world == rank (`FlatIndex.shard`). Every rank routes the full batch
identically, scans and reranks only its own leaves, and returns its local top-k. The one exchange
step of the path is a single all-gather of the [n_queries, k] (distance, id) lists per batch —
a few hundred KB, latency-bound on the fully connected xGMI links — followed by a k-way merge
that every rank performs identically (ties: distance, then rank, then position in the rank's
list; the same rule as `cuddle_merge_topk`).

Each rank selects its own top rerank_k candidates among its leaves, so up to world * rerank_k
candidates are reranked in total: recall is >= the single-index recall, and the result equals the
single-index search exactly whenever rerank_k covers every candidate.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from .flat_index import FlatIndex

PAD_ID = -1  # UINT64_MAX seen through int64


def merge_gathered(all_ids: torch.Tensor, all_dist: torch.Tensor, k: int):
    """all_ids/all_dist: [world, nq, k] (ascending per list, padded with id -1 / +inf).
    Returns (ids [nq,k], dist [nq,k], count [nq])."""
    world, nq, kk = all_dist.shape
    d = all_dist.permute(1, 0, 2).reshape(nq, world * kk)
    i = all_ids.permute(1, 0, 2).reshape(nq, world * kk)
    d = torch.where(i == PAD_ID, torch.full_like(d, float("inf")), d)
    sd, order = torch.sort(d, dim=1, stable=True)
    si = torch.gather(i, 1, order)
    sd, si = sd[:, :k].contiguous(), si[:, :k].contiguous()
    count = (si != PAD_ID).sum(dim=1).to(torch.int32)
    return si, sd, count


class ShardedSearcher:
    """search_batch over `world` leaf shards. `local_search(queries, k, probes, rerank_factor)` must
    return (ids int64 [nq,k], dist f32 [nq,k], count) tensors on the collective's device; on the
    GPU it is `DeviceIndex.search_batch_device` of this rank's shard."""

    def __init__(self, local_search: Callable, group: Optional[dist.ProcessGroup] = None):
        self.local_search = local_search
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @staticmethod
    def shard_of(flat: FlatIndex, rank: int, world: int) -> FlatIndex:
        return flat.shard(rank, world)

    def search_batch(self, queries: torch.Tensor, k: int, probes_per_level: int, rerank_factor: int):
        ids, dist_, count = self.local_search(queries, k, probes_per_level, rerank_factor)
        if self.world == 1:
            return ids, dist_, count
        nq = ids.shape[0]
        all_ids = torch.empty((self.world, nq, k), dtype=ids.dtype, device=ids.device)
        all_dist = torch.empty((self.world, nq, k), dtype=dist_.dtype, device=dist_.device)
        dist.all_gather(list(all_ids.unbind(0)), ids.contiguous(), group=self.group)
        dist.all_gather(list(all_dist.unbind(0)), dist_.contiguous(), group=self.group)
        return merge_gathered(all_ids, all_dist, k)
