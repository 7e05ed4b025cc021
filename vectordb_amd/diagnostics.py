"""Where recall is lost, stage by stage — the breakdown of the reference's recall diagnostic
(/root/reference/benches/recall_diagnostic.rs:20-146): of a query's true nearest neighbours, how many
lie in the leaves the routing reached (probe coverage), how many can be among the reranked candidates
(the reference's upper-bound estimate, :131-143), how many are in the final result."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from .flat_index import FlatIndex
from .index import SearchStats


@dataclass
class StageMetrics:
    """recall_diagnostic.rs:20-29, field for field."""
    total_vectors: int
    leaves_probed: int
    total_leaves: int
    vectors_in_probed_leaves: int
    true_nns_in_probed_leaves: int
    vectors_reranked: int
    true_nns_in_rerank_candidates: int
    true_nns_in_final_results: int


def stage_metrics(flat: FlatIndex, results: Sequence, stats: SearchStats, ground_truth: Sequence[int], k: int,
                  rerank_factor: int) -> StageMetrics:
    """diagnose_binary_search / diagnose_rabitq_search (:96-146, :148-198) for one query, from what
    search_with_stats returned. ground_truth: ids of the true nearest neighbours."""
    gt = set(int(g) for g in ground_truth)
    in_probed = set()
    for leaf in stats.leaves_searched_ids:
        a, b = int(flat.vec_offsets[leaf]), int(flat.vec_offsets[leaf + 1])
        in_probed.update(int(v) for v in flat.ids[a:b])
    nn_probed = len(gt & in_probed)
    rerank_count = min(k * rerank_factor, stats.vectors_scanned_binary)
    nn_final = len(gt & set(int(r[0]) for r in results))
    return StageMetrics(total_vectors=stats.total_vectors, leaves_probed=stats.leaves_searched, total_leaves=stats.total_leaves,
                        vectors_in_probed_leaves=len(in_probed), true_nns_in_probed_leaves=nn_probed,
                        vectors_reranked=stats.vectors_reranked_full, true_nns_in_rerank_candidates=min(nn_probed, rerank_count),
                        true_nns_in_final_results=nn_final)


def diagnose(index, queries: np.ndarray, ground_truth: np.ndarray, k: int, probes_per_level: int, rerank_factor: int) -> List[StageMetrics]:
    """The diagnostic for a batch: index = ClusteredIndex / ClusteredIndexWithRaBitQ, ground_truth [nq, >= k] ids."""
    results, stats = index.search_batch_with_stats(queries, k, probes_per_level, rerank_factor)
    return [stage_metrics(index.flat, results[i], stats[i], ground_truth[i][:k], k, rerank_factor) for i in range(len(results))]


def summarise(metrics: Sequence[StageMetrics], k: int) -> dict:
    """Averages as the reference prints them (:230-300): coverage and recall per stage, share of the index touched."""
    n = max(len(metrics), 1)
    mean = lambda f: float(sum(f(m) for m in metrics)) / n  # noqa: E731
    return {
        "queries": len(metrics),
        "leaves_probed_fraction": mean(lambda m: m.leaves_probed / max(m.total_leaves, 1)),
        "vectors_scanned_fraction": mean(lambda m: m.vectors_in_probed_leaves / max(m.total_vectors, 1)),
        "probe_coverage": mean(lambda m: m.true_nns_in_probed_leaves / k),
        "rerank_coverage_upper_bound": mean(lambda m: m.true_nns_in_rerank_candidates / k),
        "final_recall": mean(lambda m: m.true_nns_in_final_results / k),
        "lost_in_routing": mean(lambda m: (k - m.true_nns_in_probed_leaves) / k),
        "lost_in_phase1_and_rerank": mean(lambda m: (m.true_nns_in_probed_leaves - m.true_nns_in_final_results) / k),
    }
