"""stage_metrics (benches/recall_diagnostic.rs:96-146) on a hand-made index: no GPU needed."""
import numpy as np

from vectordb_amd import _ffi
from vectordb_amd.diagnostics import stage_metrics, summarise
from vectordb_amd.flat_index import FlatIndex
from vectordb_amd.index import SearchStats


def _flat():
    # root 0 with leaves 1, 2, 3 holding ids [10, 11], [12], [13, 14, 15]
    return FlatIndex(dim=2, metric=_ffi.METRIC_L2, code_kind=_ffi.CODE_BINARY, centroids=np.zeros((4, 2), np.float32),
                     child_offsets=np.array([0, 3, 3, 3, 3], np.uint32), children=np.array([1, 2, 3], np.uint32),
                     roots=np.array([0], np.uint32), vec_offsets=np.array([0, 0, 2, 3, 6], np.uint64),
                     ids=np.array([10, 11, 12, 13, 14, 15], np.uint64), vectors=np.zeros((6, 2), np.float32),
                     codes=np.zeros((6, 1), np.uint64), threshold=0.0)


def test_stage_metrics_counts():
    st = SearchStats(total_leaves=3, leaves_searched=2, leaves_searched_ids=[3, 1], total_vectors=6, vectors_scanned_binary=5,
                     vectors_reranked_full=4, tree_depth=1, probes_per_level=[1, 3, 0])
    m = stage_metrics(_flat(), [(13, 0.1), (10, 0.2)], st, [13, 12, 10], k=3, rerank_factor=2)
    assert m.vectors_in_probed_leaves == 5 and m.true_nns_in_probed_leaves == 2  # 12 lives in the leaf that was not probed
    assert m.vectors_reranked == 4 and m.true_nns_in_rerank_candidates == 2 and m.true_nns_in_final_results == 2
    assert (m.leaves_probed, m.total_leaves, m.total_vectors) == (2, 3, 6)
    s = summarise([m], 3)
    assert abs(s["probe_coverage"] - 2 / 3) < 1e-12 and abs(s["lost_in_routing"] - 1 / 3) < 1e-12
    assert s["lost_in_phase1_and_rerank"] == 0.0 and abs(s["final_recall"] - 2 / 3) < 1e-12
    # rerank budget smaller than the neighbours found: the upper bound is the budget (:131-143)
    m2 = stage_metrics(_flat(), [(13, 0.1)], st, [13, 10, 11], k=1, rerank_factor=1)
    assert m2.true_nns_in_probed_leaves == 3 and m2.true_nns_in_rerank_candidates == 1
