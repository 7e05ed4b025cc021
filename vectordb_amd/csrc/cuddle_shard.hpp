// cuddle_shard.hpp — exact phase-1 selection over leaf shards.
//
// A leaf-sharded index answers a batch with every rank routing all queries (the tree is
// replicated) and scanning only the leaves it owns. The selection of the rerank_k best candidates
// (select_top_candidates, src/index/hierarchical.rs:645-655) is a property of ALL candidates of a
// query, so the single-index result is reproduced exactly by splitting the scan kernel at the two
// points where the ranks must agree:
//   1. scan (scan_select_hamming_kernel in histogram mode): local Hamming distances -> spill
//      buffer, local (dim+1)-bin histogram -> summed over ranks by the caller;
//   2. shard_ties_kernel: threshold T and tie budget from the global histogram; number of local
//      candidates at distance T per leaf slot -> summed over ranks (a leaf has one owner, so the
//      sum is just the union);
//   3. shard_select_kernel: every rank keeps its candidates below T and, of its ties, those whose
//      position among ALL ties (leaf slot order, then in-leaf order) is below the budget.
// The union of the ranks' selections is then exactly the single-index selection, and the merge of
// the ranks' top-k by (distance, Hamming distance, leaf slot, in-leaf position) is its top-k.
//
// RaBitQ codes: the phase-1 keys are f32 estimates, too many distinct values for a histogram. Every
// rank selects its own rerank_k best as the unsharded search would (the global selection is a subset
// of the union of those), names each by a 64-bit key that orders candidates exactly as the unsharded
// selection does — (estimate, leaf slot, place in the leaf) — and the ranks all-gather the keys:
//   1. rq_exchange_keys_kernel: the rank's sorted selection -> keys;
//   2. rq_threshold_kernel: the rerank_k-th smallest of all ranks' keys (every rank computes the same);
//   3. rq_cut_kernel: the rank keeps the prefix of its selection at or below that key.
#pragma once

#include "cuddle_route.hpp"

namespace cuddle {

constexpr uint32_t kRqSlotBits = 14, kRqLeafPosBits = 18; // key = f32_key(estimate) << 32 | leaf slot << 18 | place in leaf

// thr[2q] = T, thr[2q+1] = budget; tie_count[q][slot] = local candidates of the slot at distance T
// (0 for slots past the query's leaf list). One block per query; dynamic LDS =
// scan_lds_bytes(dim, words, lcap, 0, 0).
template <class DistT>
__global__ void __launch_bounds__(256)
shard_ties_kernel(IndexDev ix, const uint32_t *leaves, uint32_t lstride, const uint32_t *lcount, uint32_t lcap,
                  const uint32_t *global_hist, uint32_t rerank_k, const void *gdist_v, uint32_t gdist_stride,
                  uint32_t *thr, uint32_t *tie_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t D = ix.dim;
    ScanLds s = scan_lds_carve(smem, D, ix.words, lcap, 0);
    const uint32_t nl = min(lcount[q], lcap);
    for (uint32_t i = tid; i <= D; i += blockDim.x) s.hist[i] = global_hist[(uint64_t)q * (D + 1) + i];
    scan_load_leaves(ix, s, leaves + (uint64_t)q * lstride, nl, 0, nullptr); // barriers inside: hist visible after
    if (w == 0) { // smallest T with #(<= T) >= rerank_k over ALL shards; none: every candidate is selected
        uint32_t carry = 0, T = D + 1, budget = 0;
        bool done = false;
        for (uint32_t b0 = 0; b0 <= D && !done; b0 += 64) {
            const uint32_t b = b0 + lane;
            const uint32_t v = b <= D ? s.hist[b] : 0;
            const uint32_t incl = wave_inclusive_sum(v) + carry;
            const uint64_t hit = __ballot(incl >= rerank_k);
            if (hit) {
                const uint32_t first = __ffsll((unsigned long long)hit) - 1;
                const uint32_t incl_f = (uint32_t)__builtin_amdgcn_readlane((int)incl, first), v_f = (uint32_t)__builtin_amdgcn_readlane((int)v, first);
                T = b0 + first;
                budget = rerank_k - (incl_f - v_f);
                done = true;
            }
            carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) {
            s.misc[0] = T;
            s.misc[1] = budget;
            thr[2 * q] = T;
            thr[2 * q + 1] = budget;
        }
    }
    __syncthreads();
    const uint32_t T = s.misc[0];
    const DistT *dist = reinterpret_cast<const DistT *>(gdist_v) + (uint64_t)q * gdist_stride;
    for (uint32_t sl = w; sl < lstride; sl += nw) {
        uint32_t cnt = 0;
        if (sl < nl) {
            const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                cnt += __popcll(__ballot(i < n && (uint32_t)dist[base + (i < n ? i : 0)] == T));
            }
        }
        if (lane == 0) tie_count[(uint64_t)q * lstride + sl] = cnt;
    }
}

// The rank's share of the selection, in scan order: sel_pos (leaf-contiguous position), sel_key
// (Hamming distance), sel_ord (the candidate's place in the global phase-1 order:
// Hamming << 48 | leaf slot << 24 | in-leaf position). One block per query; same LDS as above.
template <class DistT>
__global__ void __launch_bounds__(256)
shard_select_kernel(IndexDev ix, const uint32_t *leaves, uint32_t lstride, const uint32_t *lcount, uint32_t lcap,
                    const uint32_t *thr, const uint32_t *global_ties, const void *gdist_v, uint32_t gdist_stride,
                    uint32_t *sel_pos, uint32_t *sel_key, uint64_t *sel_ord, uint32_t sel_stride, uint32_t *sel_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    ScanLds s = scan_lds_carve(smem, ix.dim, ix.words, lcap, 0);
    const uint32_t nl = min(lcount[q], lcap);
    const uint32_t T = thr[2 * q], budget = thr[2 * q + 1];
    scan_load_leaves(ix, s, leaves + (uint64_t)q * lstride, nl, 0, nullptr);
    // ties before each slot, over all shards (s.ltile is free: no tiles are cut here)
    __syncthreads();
    for (uint32_t sl = tid; sl < nl; sl += blockDim.x) s.ltile[sl] = global_ties[(uint64_t)q * lstride + sl];
    __syncthreads();
    wave0_exclusive_scan_lds(s.ltile, nl);
    __syncthreads();
    const DistT *dist = reinterpret_cast<const DistT *>(gdist_v) + (uint64_t)q * gdist_stride;
    uint32_t *sel_off = s.lcode; // [lcap + 1]; codes are not read here: the slot's first output rank, total at [nl]
    auto quota_of = [&](uint32_t sl) -> uint32_t { // how many of the slot's ties are inside the budget
        const uint32_t before = s.ltile[sl], ties = s.ltile[sl + 1] - before;
        return before >= budget ? 0u : min(ties, budget - before);
    };
    for (uint32_t sl = w; sl < nl; sl += nw) {
        const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
        uint32_t lt = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            lt += __popcll(__ballot(i < n && (uint32_t)dist[base + (i < n ? i : 0)] < T));
        }
        if (lane == 0) sel_off[sl] = n ? lt + quota_of(sl) : 0u; // a leaf has one owner: its ties are all here or all elsewhere
    }
    __syncthreads();
    wave0_exclusive_scan_lds(sel_off, nl);
    __syncthreads();
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    uint64_t *so = sel_ord + (uint64_t)q * sel_stride;
    const uint64_t lm = lanemask_lt();
    for (uint32_t sl = w; sl < nl; sl += nw) {
        const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
        if (n == 0) continue;
        const uint32_t quota = quota_of(sl);
        uint32_t out = sel_off[sl], tie_run = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool in = i < n;
            const uint32_t kv = in ? (uint32_t)dist[base + i] : 0xFFFFFFFFu;
            const bool tie = in && kv == T;
            const uint64_t tb = __ballot(tie);
            const bool flag = in && (kv < T || (tie && tie_run + (uint32_t)__popcll(tb & lm) < quota));
            const uint64_t fb = __ballot(flag);
            if (flag) {
                const uint32_t r = out + (uint32_t)__popcll(fb & lm);
                if (r < sel_stride) {
                    sp[r] = s.lbeg[sl] + i;
                    sk[r] = kv;
                    so[r] = ((uint64_t)kv << 48) | ((uint64_t)sl << 24) | i;
                }
            }
            tie_run += __popcll(tb);
            out += __popcll(fb);
        }
    }
    if (tid == 0) sel_count[q] = min(sel_off[nl], sel_stride);
}

// Merge of the shards' result lists (cuddle_shard_finish: ascending, padded with id = ~0) into the
// global top-k per query, ascending by (distance, order): the finalize_results order of the unsharded
// index (hierarchical.rs:678-691). List l of an array ([nq][k]) starts that array's stride (bytes) after
// list l - 1; n_lists * k <= 64 * KPL.
// One wave per query, entries in registers, k rounds of "smallest (distance, order) of the wave".
template <int KPL>
__global__ void __launch_bounds__(256)
merge_ordered_kernel(const uint64_t *ids, const float *dist, const uint64_t *order, uint64_t ids_stride,
                     uint64_t dist_stride, uint64_t order_stride, uint32_t n_lists,
                     uint32_t nq, uint32_t k, uint64_t *out_ids, float *out_dist, uint32_t *out_count) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    auto at = [&](const auto *base, uint64_t list_stride, uint32_t e) { // entry e = (list, rank in list) of this query
        typedef decltype(base) P;
        return reinterpret_cast<P>(reinterpret_cast<const char *>(base) + (uint64_t)(e / k) * list_stride) + (uint64_t)q * k + e % k;
    };
    const uint32_t total = n_lists * k;
    uint32_t dk[KPL];
    uint64_t od[KPL];
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
        const uint32_t e = t * 64 + lane;
        dk[t] = 0xFFFFFFFFu;
        od[t] = ~0ull;
        if (e < total) {
            if (*at(ids, ids_stride, e) != ~0ull) {
                dk[t] = f32_key(*at(dist, dist_stride, e) + 0.0f); // -0 orders as +0, as the float comparison does
                od[t] = *at(order, order_stride, e);
            }
        }
    }
    uint32_t n = 0;
    for (uint32_t r = 0; r < k; ++r) {
        uint32_t bd = dk[0], bt = 0;
        uint64_t bo = od[0];
#pragma unroll
        for (int t = 1; t < KPL; ++t)
            if (dk[t] < bd || (dk[t] == bd && od[t] < bo)) {
                bd = dk[t];
                bo = od[t];
                bt = t;
            }
        const uint32_t md = wave_min_u32(bd);
        const uint64_t mo = wave_min_u64(bd == md ? bo : ~0ull);
        if (md == 0xFFFFFFFFu && mo == ~0ull) break; // only padding left
        const bool win = bd == md && bo == mo;       // order keys are unique: one lane
        const uint64_t wb = __ballot(win);
        const uint32_t src = (uint32_t)__ffsll((unsigned long long)wb) - 1;
        if (win && lane == src) { // the r-th result, written by the lane that holds it
            const uint32_t e_w = bt * 64 + lane;
            out_ids[(uint64_t)q * k + r] = *at(ids, ids_stride, e_w);
            out_dist[(uint64_t)q * k + r] = *at(dist, dist_stride, e_w);
#pragma unroll
            for (int t = 0; t < KPL; ++t)
                if ((uint32_t)t == bt) {
                    dk[t] = 0xFFFFFFFFu;
                    od[t] = ~0ull;
                }
        }
        ++n;
    }
    for (uint32_t i = n + lane; i < k; i += 64) {
        out_ids[(uint64_t)q * k + i] = ~0ull;
        out_dist[(uint64_t)q * k + i] = __builtin_inff();
    }
    if (lane == 0 && out_count) out_count[q] = n;
}

// keys[q][i] (i < rerank_k; ~0 past the rank's selection) and sel_ord[q][i] = the exchange key of the i-th
// entry of the rank's sorted phase-1 selection. The leaf slot of an entry is found from its position:
// the query's leaves own disjoint position ranges (leaves of other ranks are empty here), so the slots'
// first positions are sorted once per query and every entry's slot is a binary search away.
// One block per query; dynamic LDS = rq_exchange_lds_bytes(lcap).
inline size_t rq_exchange_lds_bytes(uint32_t lcap) {
    uint32_t npad = 1;
    while (npad < lcap) npad <<= 1;
    return (size_t)npad * 8;
}

__global__ void __launch_bounds__(256)
rq_exchange_keys_kernel(IndexDev ix, const uint32_t *leaves, uint32_t lstride, const uint32_t *lcount, uint32_t lcap,
                        const uint32_t *sel_pos, const uint32_t *sel_key, uint32_t sel_stride, const uint32_t *sel_count,
                        uint32_t rerank_k, uint64_t *keys, uint64_t *sel_ord) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *first = reinterpret_cast<uint64_t *>(smem); // [npad] (first position << 32 | slot) of the non-empty slots, ascending
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t nl = min(lcount[q], lcap), n = min(min(sel_count[q], sel_stride), rerank_k);
    const uint32_t npad = next_pow2(max(lcap, 1u));
    for (uint32_t sl = tid; sl < npad; sl += blockDim.x) {
        uint64_t e = ~0ull;
        if (sl < nl) {
            const uint32_t node = leaves[(uint64_t)q * lstride + sl];
            const uint32_t b = (uint32_t)ix.vec_off[node], en = (uint32_t)ix.vec_off[node + 1];
            if (en > b) e = ((uint64_t)b << 32) | sl;
        }
        first[sl] = e;
    }
    __syncthreads();
    block_bitonic_sort_u64(first, npad);
    for (uint32_t i = tid; i < rerank_k; i += blockDim.x) {
        uint64_t key = ~0ull;
        if (i < n) {
            const uint32_t pos = sel_pos[(uint64_t)q * sel_stride + i];
            uint32_t lo = 0, hi = npad; // the last entry whose first position is <= pos (a selected position lies in a non-empty slot)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)(first[mid] >> 32) <= pos) lo = mid; else hi = mid;
            }
            const uint64_t e = first[lo];
            key = ((uint64_t)f32_key(__uint_as_float(sel_key[(uint64_t)q * sel_stride + i])) << 32) |
                  ((uint64_t)(uint32_t)e << kRqLeafPosBits) | (pos - (uint32_t)(e >> 32));
            sel_ord[(uint64_t)q * sel_stride + i] = key;
        }
        keys[(uint64_t)q * rerank_k + i] = key;
    }
}

// thr[q] = the rerank_k-th smallest of the query's keys over all ranks (all_keys: [world][nq][rerank_k]); with
// fewer than rerank_k candidates in all, every one of them is selected (thr = ~0 - 1; ~0 is padding).
// Keys are distinct, so the selection is exactly the keys <= thr[q]. Two radix selections: the high
// word, then the low word among the keys that share the threshold's high word. One block per query.
__global__ void __launch_bounds__(256)
rq_threshold_kernel(const uint64_t *all_keys, uint32_t world, uint32_t nq, uint32_t rerank_k, uint64_t *thr) {
    __shared__ uint32_t hist[256], misc[2], wred[8], s_valid;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, n = world * rerank_k;
    auto key_at = [&](uint32_t i) { return all_keys[((uint64_t)(i / rerank_k) * nq + q) * rerank_k + i % rerank_k]; };
    if (tid == 0) s_valid = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n; i += blockDim.x) mine += key_at(i) != ~0ull ? 1u : 0u;
    if (mine) atomicAdd(&s_valid, mine);
    __syncthreads();
    if (s_valid <= rerank_k) {
        if (tid == 0) thr[q] = ~0ull - 1;
        return;
    }
    uint32_t budget;
    const uint32_t hi = block_radix_select_aligned(n, rerank_k, [&](uint32_t i) { return (uint32_t)(key_at(i) >> 32); }, hist, misc,
                                                   wred, budget);
    __syncthreads();
    uint32_t b2;
    const uint32_t lo = block_radix_select_aligned(
        n, budget, [&](uint32_t i) { const uint64_t k = key_at(i); return (uint32_t)(k >> 32) == hi ? (uint32_t)k : 0xFFFFFFFFu; },
        hist, misc, wred, b2);
    if (tid == 0) thr[q] = ((uint64_t)hi << 32) | lo;
}

// The rank's share of the global selection: its own selection is sorted by the same keys, so the share is
// a prefix. sel_count[q] = number of the rank's keys <= thr[q].
__global__ void __launch_bounds__(256)
rq_cut_kernel(const uint64_t *sel_ord, uint32_t sel_stride, uint32_t *sel_count, uint32_t rerank_k, const uint64_t *thr,
              uint32_t nq) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t n = min(min(sel_count[q], sel_stride), rerank_k);
    const uint64_t t = thr[q], *k = sel_ord + (uint64_t)q * sel_stride;
    uint32_t lo = 0, hi = n; // keys [0, lo) <= t < keys [hi, n)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (k[mid] <= t) lo = mid + 1; else hi = mid;
    }
    sel_count[q] = lo;
}

} // namespace cuddle
