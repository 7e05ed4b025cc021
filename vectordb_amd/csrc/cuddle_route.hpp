// cuddle_route.hpp — centroid routing kernels (reference: src/index/hierarchical.rs:349-423).
//
// Two ways to evaluate one tree level for a batch of queries, with bit-identical results:
//
//  * exact path (any level): route_dist_kernel computes every query->centroid distance in the
//    arithmetic the reference uses for that frontier size, route_select_kernel picks the nodes.
//
//  * MFMA path (a level whose frontier is the same node list for every query and wider than
//    probes_per_level): the query x centroid contraction runs on the bf16 matrix cores
//    (v_mfma_f32_32x32x16_bf16 over operands split into bf16 hi and lo parts, f32 accumulation)
//    with precomputed squared norms and gives an *approximate* distance d~ for every (query, node); the probes + margin smallest d~ per query are then re-evaluated
//    exactly, sorted, and accepted only if the error bound proves that no unselected node can
//    belong to the exact top `probes`. Queries that fail the proof are re-routed by the exact path.
#pragma once

#include "cuddle_kernels.hpp"

namespace cuddle {

// ---------------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ float f32_from_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Radix select over n keys (key(i) -> u32): returns the value T of the want-th smallest key
// (1-based) and, in `budget`, how many of the keys equal to T belong to the `want` smallest.
// hist: 256 u32 of LDS, misc: 2 u32 of LDS. All threads of the block call.
template <class KeyFn>
__device__ __forceinline__ uint32_t block_radix_select(uint32_t n, uint32_t want, KeyFn key, uint32_t *hist,
                                                       uint32_t *misc, uint32_t &budget) {
    uint32_t prefix = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const uint32_t himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            uint32_t kv = key(i);
            if ((kv & himask) == prefix) atomicAdd(&hist[(kv >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) { // wave 0: prefix-sum the 256 buckets, 4 per lane
            const uint32_t l = threadIdx.x;
            const uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const uint32_t s4 = h0 + h1 + h2 + h3;
            const uint32_t incl = wave_inclusive_sum(s4);
            const uint64_t hit = __ballot(incl >= want);
            const uint32_t first = hit ? (uint32_t)__ffsll((unsigned long long)hit) - 1 : 63u;
            if (l == first) {
                uint32_t cum = incl - s4, b = 4 * l;
                if (cum + h0 < want) { cum += h0; ++b;
                    if (cum + h1 < want) { cum += h1; ++b;
                        if (cum + h2 < want) { cum += h2; ++b; } } }
                misc[0] = prefix | (b << shift);
                misc[1] = want - cum;
            }
        }
        __syncthreads();
        prefix = misc[0];
        want = misc[1];
        __syncthreads();
    }
    budget = want;
    return prefix;
}

// block_radix_select whose digits start at the highest bit in which the block's smallest and largest key
// differ: keys that share their leading bits (floats of one magnitude) would otherwise all count into one
// bucket of the first passes, one same-address LDS atomic after another. wred: 2 * (blockDim.x / 64) u32.
template <class KeyFn>
__device__ __forceinline__ uint32_t block_radix_select_aligned(uint32_t n, uint32_t want, KeyFn key, uint32_t *hist,
                                                               uint32_t *misc, uint32_t *wred, uint32_t &budget) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint32_t kv = key(i);
        kmin = min(kmin, kv);
        kmax = max(kmax, kv);
    }
    kmin = wave_min_u32(kmin);
    kmax = ~wave_min_u32(~kmax);
    __syncthreads(); // wred may still be read by an earlier call
    if (lane == 0) {
        wred[w] = kmin;
        wred[nw + w] = kmax;
    }
    __syncthreads();
    for (uint32_t v = 0; v < nw; ++v) {
        kmin = min(kmin, wred[v]);
        kmax = max(kmax, wred[nw + v]);
    }
    const uint32_t diff = kmin ^ kmax;
    if (diff == 0) { // all keys equal (or n == 0)
        budget = want;
        return kmin;
    }
    uint32_t rem = 32 - __clz(diff); // bits still undecided
    uint32_t prefix = rem == 32 ? 0u : (kmin >> rem) << rem;
    while (rem > 0) {
        const uint32_t wbits = min(8u, rem), shift = rem - wbits, dmask = (1u << wbits) - 1;
        for (uint32_t i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += blockDim.x) {
            const uint32_t kv = key(i);
            if (rem == 32 || (kv >> rem) == (prefix >> rem)) atomicAdd(&hist[(kv >> shift) & dmask], 1u);
        }
        __syncthreads();
        if (tid < 64) { // wave 0: prefix-sum the 256 buckets, 4 per lane
            const uint32_t l = tid;
            const uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const uint32_t s4 = h0 + h1 + h2 + h3;
            const uint32_t incl = wave_inclusive_sum(s4);
            const uint64_t hit = __ballot(incl >= want);
            const uint32_t first = hit ? (uint32_t)__ffsll((unsigned long long)hit) - 1 : 63u;
            if (l == first) {
                uint32_t cum = incl - s4, b = 4 * l;
                if (cum + h0 < want) { cum += h0; ++b;
                    if (cum + h1 < want) { cum += h1; ++b;
                        if (cum + h2 < want) { cum += h2; ++b; } } }
                misc[0] = prefix | (b << shift);
                misc[1] = want - cum;
            }
        }
        __syncthreads();
        prefix = misc[0];
        want = misc[1];
        rem = shift;
        __syncthreads();
    }
    budget = want;
    return prefix;
}

// Optional output of route_expand for the next level's grouped distance kernel: one (query, chosen
// internal node) pair per selection slot, and the number of pairs per parent node.
//   info[q * pstride + i]   = (q << 20) | position of the node's first child in the next frontier,
//                             or 0xFFFFFFFF (slot unused, leaf, or query not grouped)
//   parent[q * pstride + i] = level position of the node
//   cnt[level position]    += 1 per pair
//   small[q]                = 1 when the query's next frontier has at most min_n nodes (its distances
//                             use the small-frontier arithmetic and are left to route_dist_kernel)
struct PairOut {
    uint32_t *info = nullptr;
    uint32_t *parent = nullptr;
    uint32_t *slot = nullptr; // the pair's place among its parent's pairs (the value cnt had when it was counted)
    uint32_t *cnt = nullptr;
    uint32_t *small = nullptr;
    uint32_t pstride = 0;
    uint32_t min_n = 0;
    // A level of few parents takes a pair from nearly every query for every parent: thousands of returning atomic adds
    // on each of a hundred counters, one after the other. Such a level counts into 2^shift counters per parent instead
    // (counter = parent << shift | query & (2^shift - 1)); the pair machinery then sees 2^shift parents for every
    // real one, adjacent, so that a parent's pairs still lie together (pair_shift_of, pair_scan_kernel).
    uint32_t shift = 0;
    // seg != null: a grouped query's next frontier is not written out node by node; the query gets its
    // segment table instead (seg + q * seg_stride(pstride)): start[0..pstride] = position of each chosen
    // node's first child in the frontier (entries past the selection = the frontier's size), then
    // cbase[0..pstride) = where the node's children start in the index's child array. frontier_node()
    // maps a frontier position back to its node. Queries that are not grouped still get their frontier.
    uint32_t *seg = nullptr;
};
__host__ __device__ __forceinline__ uint32_t seg_stride(uint32_t pstride) { return 2 * pstride + 2; }

// Node at position `pos` of the frontier a segment table stands for (st: the query's table, LDS or global).
__device__ __forceinline__ uint32_t frontier_node(const IndexDev &ix, const uint32_t *st, uint32_t pstride, uint32_t pos) {
    uint32_t lo = 0, hi = pstride; // start[lo] <= pos < start[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (st[mid] <= pos) lo = mid; else hi = mid;
    }
    return ix.children[st[pstride + 1 + lo] + (pos - st[lo])];
}
constexpr uint32_t kPairOffBits = 20; // next-frontier positions below 2^20, queries per chunk below 2^12
constexpr int kGroupNQ = 8;           // queries per tile of route_dist_grouped_kernel
constexpr int kGroupPairs = 64;       // pairs per block of route_dist_grouped_kernel (8 tiles)
typedef float v2f __attribute__((ext_vector_type(2)));

// Radix select over keys held in registers (KPT per thread, key t of thread tid is item
// t * THREADS + tid; items >= n hold 0xFFFFFFFF): returns the want-th smallest key (1-based) and,
// in `budget`, how many of the keys equal to it belong to the `want` smallest. The leading bits
// that every key shares (found from the block's min and max) cost nothing; each pass over the rest
// is one counting sweep with LDS atomics and ONE barrier: each wave then scans the 256 counters itself.
// hist: 4 * 256 u32 of LDS, zeroed here; wred: 2 * (THREADS / 64) u32. All threads call.
// UPPER: stop at the first digit whose bucket holds at most four keys and return the largest key that bucket can
// hold - an upper bound of the want-th smallest key that at least `want` and at most `want` + 3 keys reach, for
// callers that only need a cut (budget is not set).
template <int THREADS, int KPT, bool UPPER = false>
__device__ __forceinline__ uint32_t reg_radix_select(const uint32_t (&keys)[KPT], uint32_t n, uint32_t want, uint32_t *hist,
                                                     uint32_t *wred, uint32_t &budget) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int t = 0; t < KPT; ++t) {
        if ((uint32_t)t * THREADS + tid < n) {
            kmin = min(kmin, keys[t]);
            kmax = max(kmax, keys[t]);
        }
    }
    kmin = wave_min_u32(kmin);
    kmax = ~wave_min_u32(~kmax);
    if (lane == 0) {
        wred[w] = kmin;
        wred[THREADS / 64 + w] = kmax;
    }
    for (uint32_t i = tid; i < 4 * 256; i += THREADS) hist[i] = 0;
    __syncthreads();
    for (uint32_t v = 0; v < THREADS / 64; ++v) {
        kmin = min(kmin, wred[v]);
        kmax = max(kmax, wred[THREADS / 64 + v]);
    }
    // Bits above the highest bit in which min and max differ are common to every key; the digits
    // start right below them, so that the first (all-keys) pass spreads over the whole histogram.
    const uint32_t diff = kmin ^ kmax;
    if (diff == 0) { // all keys equal
        budget = want;
        return kmin;
    }
    uint32_t rem = 32 - __clz(diff);                                  // bits still undecided
    uint32_t prefix = rem == 32 ? 0u : (kmin >> rem) << rem;
    for (int pass = 0; rem > 0; ++pass) {
        const uint32_t wbits = min(8u, rem), shift = rem - wbits, dmask = (1u << wbits) - 1;
        uint32_t *h = hist + 256 * pass;
#pragma unroll
        for (int t = 0; t < KPT; ++t) {
            const bool in = (uint32_t)t * THREADS + tid < n && (rem == 32 || (keys[t] >> rem) == (prefix >> rem));
            if (in) atomicAdd(&h[(keys[t] >> shift) & dmask], 1u);
        }
        __syncthreads();
        // every wave: prefix-sum the 256 counters, 4 per lane, and find the bucket holding the want-th key
        const uint32_t h0 = h[4 * lane], h1 = h[4 * lane + 1], h2 = h[4 * lane + 2], h3 = h[4 * lane + 3];
        const uint32_t s4 = h0 + h1 + h2 + h3;
        const uint32_t incl = wave_inclusive_sum(s4);
        const uint64_t hit = __ballot(incl >= want);
        const uint32_t first = hit ? (uint32_t)__ffsll((unsigned long long)hit) - 1 : 63u;
        uint32_t cum = incl - s4, b = 4 * lane, hb = h0;
        if (cum + h0 < want) { cum += h0; ++b; hb = h1;
            if (cum + h1 < want) { cum += h1; ++b; hb = h2;
                if (cum + h2 < want) { cum += h2; ++b; hb = h3; } } }
        prefix |= (uint32_t)__builtin_amdgcn_readlane((int)b, first) << shift;
        want -= (uint32_t)__builtin_amdgcn_readlane((int)cum, first);
        rem = shift;
        // UPPER: the bucket's upper edge will do once the bucket holds only a few keys (they are all the cut lets in
        // beyond the want-th)
        if (UPPER && (uint32_t)__builtin_amdgcn_readlane((int)hb, first) <= 4) return prefix | (rem ? (1u << rem) - 1 : 0u);
    }
    budget = want;
    return prefix;
}

// Split the chosen nodes into leaves and internal nodes (hierarchical.rs:411-413), append the
// leaves to the query's leaf list (:416) and write the children of the internal nodes, in order,
// as the next frontier (:419-422; skipped when nf is null: the next level's frontier is then the
// level's full node list, known without writing it per query). skeys[i] (low 32 bits) = position
// in `fr` of the i-th chosen node. xs: 4 * cap1 u32 of LDS scratch, cap1 > nsel. All threads call.
template <class NodeFn>
__device__ __forceinline__ void route_expand_fn(const IndexDev &ix, NodeFn node_at, const uint64_t *skeys,
                                                uint32_t nsel, uint32_t *xs, uint32_t cap1, uint32_t *nf,
                                                uint32_t fstride, uint32_t *next_fcount_q, uint32_t *lv, uint32_t lstride,
                                                uint32_t *lcount_q, const PairOut &po = PairOut(), uint32_t q = 0) {
    uint32_t *cnt_leaf = xs, *cnt_child = xs + cap1, *sel_node = xs + 2 * cap1, *sel_cb = xs + 3 * cap1;
    const uint32_t tid = threadIdx.x;
    const uint32_t lbase = *lcount_q; // read by every thread three barriers before thread 0 replaces it
    for (uint32_t i = tid; i < nsel; i += blockDim.x) {
        const uint32_t node = node_at((uint32_t)skeys[i]);
        const uint32_t cb = ix.child_off[node], cc = ix.child_off[node + 1] - cb;
        cnt_leaf[i] = cc == 0 ? 1u : 0u;
        cnt_child[i] = cc;
        sel_node[i] = node;
        sel_cb[i] = cb;
    }
    __syncthreads();
    wave0_exclusive_scan_lds(cnt_leaf, nsel);
    __syncthreads();
    wave0_exclusive_scan_lds(cnt_child, nsel);
    __syncthreads();
    if (tid == 0) {
        *lcount_q = min(lbase + cnt_leaf[nsel], lstride);
        if (nf) *next_fcount_q = min(cnt_child[nsel], fstride);
    }
    const uint32_t nchild = cnt_child[nsel];
    const bool grouped = po.info && nf && nchild > po.min_n && nchild <= fstride;
    if (po.info) {
        for (uint32_t i = tid; i < po.pstride; i += blockDim.x) {
            uint32_t info = 0xFFFFFFFFu, par = 0, slot = 0;
            if (grouped && i < nsel && cnt_child[i + 1] > cnt_child[i]) {
                par = (ix.level_pos[sel_node[i]] << po.shift) | (q & ((1u << po.shift) - 1));
                info = (q << kPairOffBits) | cnt_child[i];
                slot = atomicAdd(&po.cnt[par], 1u);
            }
            po.info[(uint64_t)q * po.pstride + i] = info;
            po.parent[(uint64_t)q * po.pstride + i] = par;
            po.slot[(uint64_t)q * po.pstride + i] = slot;
        }
        if (tid == 0) po.small[q] = grouped ? 0u : 1u;
    }
    const bool seg_only = grouped && po.seg;
    if (seg_only) {
        uint32_t *st = po.seg + (uint64_t)q * seg_stride(po.pstride);
        for (uint32_t i = tid; i <= po.pstride; i += blockDim.x) st[i] = cnt_child[min(i, nsel)];
        for (uint32_t i = tid; i < po.pstride; i += blockDim.x) st[po.pstride + 1 + i] = i < nsel ? sel_cb[i] : 0u;
    }
    // leaves are appended by the thread of their selection slot; the children are copied one output
    // position per thread (the owning node found by bisection of the prefix sums in LDS), so that a
    // selection of many small families costs as little as one of a few large ones
    for (uint32_t i = tid; i < nsel; i += blockDim.x)
        if (cnt_child[i + 1] == cnt_child[i]) {
            const uint32_t slot = lbase + cnt_leaf[i];
            if (slot < lstride) lv[slot] = sel_node[i];
        }
    if (nf && !seg_only) {
        const uint32_t total = min(cnt_child[nsel], fstride);
        for (uint32_t o = tid; o < total; o += blockDim.x) {
            uint32_t lo = 0, hi = nsel; // cnt_child[lo] <= o < cnt_child[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cnt_child[mid] <= o) lo = mid; else hi = mid;
            }
            nf[o] = ix.children[sel_cb[lo] + (o - cnt_child[lo])];
        }
    }
}

__device__ __forceinline__ void route_expand(const IndexDev &ix, const uint32_t *fr, const uint64_t *skeys,
                                             uint32_t nsel, uint32_t *xs, uint32_t cap1, uint32_t *nf,
                                             uint32_t fstride, uint32_t *next_fcount_q, uint32_t *lv, uint32_t lstride,
                                             uint32_t *lcount_q, const PairOut &po = PairOut(), uint32_t q = 0) {
    route_expand_fn(ix, [&](uint32_t pos) { return fr[pos]; }, skeys, nsel, xs, cap1, nf, fstride, next_fcount_q, lv, lstride,
                    lcount_q, po, q);
}

// The frontier of the queries with flag[q] != 0 that only have a segment table (small[q] == 0), written
// out node by node for the kernels that walk whole frontiers. One block per query.
__global__ void __launch_bounds__(256)
route_materialise_kernel(IndexDev ix, const uint32_t *seg, uint32_t pstride, const uint32_t *flag, const uint32_t *small,
                         uint32_t *frontier, uint32_t fstride, const uint32_t *fcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *st = reinterpret_cast<uint32_t *>(smem);
    const uint32_t q = blockIdx.x;
    if (!flag[q] || small[q]) return;
    const uint32_t *sg = seg + (uint64_t)q * seg_stride(pstride);
    for (uint32_t i = threadIdx.x; i < seg_stride(pstride); i += blockDim.x) st[i] = sg[i];
    __syncthreads();
    const uint32_t n = min(fcount[q], fstride);
    for (uint32_t o = threadIdx.x; o < n; o += blockDim.x) frontier[(uint64_t)q * fstride + o] = frontier_node(ix, st, pstride, o);
}

// Level start: frontier = root_ids for every query (hierarchical.rs:346). write_roots = 0 when the
// first level is evaluated from the shared root list.
// zero != null: the first block also clears zero[0..n_zero) (a search's counters, instead of a memset launch of its own).
__device__ __forceinline__ void route_init_block(uint32_t bx, uint32_t q, const uint32_t *roots, uint32_t n_roots,
                                                 uint32_t *frontier, uint32_t fstride, uint32_t *fcount, uint32_t *lcount,
                                                 uint32_t nq, uint32_t write_roots, uint32_t *zero, uint32_t n_zero) {
    uint32_t i = bx * blockDim.x + threadIdx.x;
    if (zero && q == 0 && bx == 0)
        for (uint32_t z = threadIdx.x; z < n_zero; z += blockDim.x) zero[z] = 0;
    if (q >= nq) return;
    if (write_roots && i < n_roots) frontier[(uint64_t)q * fstride + i] = roots[i];
    if (i == 0) {
        fcount[q] = n_roots;
        lcount[q] = 0;
    }
}

__global__ void route_init_kernel(const uint32_t *roots, uint32_t n_roots, uint32_t *frontier, uint32_t fstride,
                                  uint32_t *fcount, uint32_t *lcount, uint32_t nq, uint32_t write_roots,
                                  uint32_t *zero = nullptr, uint32_t n_zero = 0) {
    route_init_block(blockIdx.x, blockIdx.y, roots, n_roots, frontier, fstride, fcount, lcount, nq, write_roots, zero, n_zero);
}

// A search's first two steps in one launch: blocks [0, qblocks) are binary_quantize_kernel's, the rest route_init_kernel's
// (block qblocks + q * rblocks + bx does (bx, q)). Block 256.
__global__ void __launch_bounds__(256)
quantize_init_kernel(const float *v, uint64_t n, uint32_t dim, uint32_t words, float thr, uint64_t *codes, uint32_t qblocks,
                     uint32_t rblocks, const uint32_t *roots, uint32_t n_roots, uint32_t *frontier, uint32_t fstride,
                     uint32_t *fcount, uint32_t *lcount, uint32_t nq, uint32_t write_roots, uint32_t *zero, uint32_t n_zero) {
    if (blockIdx.x < qblocks) {
        binary_quantize_block(blockIdx.x, v, n, dim, words, thr, codes);
        return;
    }
    const uint32_t r = blockIdx.x - qblocks;
    route_init_block(r % rblocks, r / rblocks, roots, n_roots, frontier, fstride, fcount, lcount, nq, write_roots, zero, n_zero);
}

// ---------------------------------------------------------------------------------------------
// exact path
// ---------------------------------------------------------------------------------------------

// Exact distances query -> frontier centroids, in the arithmetic search() uses for that
// frontier size (hierarchical.rs:355-391): n > 100 or n < 4: distance(metric) per node;
// 4 <= n <= 100: l2_squared_batch, i.e. batch4 arithmetic for the first 4*floor(n/4) nodes and
// l2_squared for the rest, whatever the metric. stats_variant: always distance(metric) (:868-886).
// Nothing is computed for a query whose nodes are all kept (n <= probes, :395-397).
// only_flag != null: only queries with only_flag[q] != 0 are processed. shared_fr != null: every
// query's frontier is the list shared_fr[0..shared_n) (frontier / fcount are not read).
// grid = (gx, nq) with any gx >= 1 (blocks stride over the nodes), block = 256: one 4-lane group
// per (query, node).
// The work of one block of route_dist_kernel for query q: block bx of gx over the frontier's nodes. qs: dim floats of
// LDS. All threads of the block call (one barrier inside).
__device__ __forceinline__ void route_dist_block(const IndexDev &ix, float *qs, const float *qg, const uint32_t *fr, float *fd,
                                                 uint32_t n, uint32_t probes, uint32_t stats_variant, float qnorm_q, uint32_t bx,
                                                 uint32_t gx) {
    const uint32_t D = ix.dim;
    const bool small = !stats_variant && n >= 4 && n <= 100;
    if (small ? (bx != 0) : (bx * (blockDim.x >> 2) >= n)) return; // (block-uniform)
    for (uint32_t i = threadIdx.x; i < D; i += blockDim.x) qs[i] = qg[i];
    __syncthreads();
    if (small) {
        uint32_t j = threadIdx.x;
        if (j < n) {
            const float *c = ix.centroids + (uint64_t)fr[j] * D;
            fd[j] = (j < (n & ~3u)) ? l2_batch4_serial(qs, c, D) : neon16_serial<true>(qs, c, D);
        }
        return;
    }
    const uint32_t sub = threadIdx.x & 3, groups = blockDim.x >> 2;
    const bool vec = (D & 3) == 0;
    for (uint32_t j0 = bx * groups; j0 < n; j0 += gx * groups) { // block-uniform trip count
        const uint32_t j = j0 + (threadIdx.x >> 2);
        const bool act = j < n;
        const uint32_t node = fr[act ? j : 0];
        float d = distance_group4_qn(qs, ix.centroids + (uint64_t)node * D, D, ix.metric, sub, vec, qnorm_q);
        if (act && sub == 0) fd[j] = d;
    }
}

__global__ void __launch_bounds__(256)
route_dist_kernel(IndexDev ix, const float *queries, const uint32_t *frontier, uint32_t fstride,
                  const uint32_t *fcount, const uint32_t *shared_fr, uint32_t shared_n, float *fdist, uint32_t probes,
                  uint32_t stats_variant, const uint32_t *only_flag, const float *qnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.y;
    if (only_flag && !only_flag[q]) return;
    const uint32_t n = shared_fr ? shared_n : fcount[q];
    if (!stats_variant && n <= probes) return;
    route_dist_block(ix, reinterpret_cast<float *>(smem), queries + (uint64_t)q * ix.dim,
                     shared_fr ? shared_fr : frontier + (uint64_t)q * fstride, fdist + (uint64_t)q * fstride, n, probes,
                     stats_variant, ix.metric == kMetricCosine ? qnorm[q] : 0.f, blockIdx.x, gridDim.x);
}

// route_dist_kernel for a level whose frontier is shared by every query and takes the batch arithmetic (4 <= n <= 100
// nodes: the roots of a tree searched with fewer probes than it has roots). With a block per query every block would
// read the same n centroids from L2 (50 KB for 100 x 128, past what a CU's L1 keeps: 4096 queries were 60 us of L2
// reads); here a block of 16 waves stages them in LDS once and each wave serves one query at a time, a 4-lane group per
// node. Rows are `cstride` floats apart, cstride = 4 (mod 64) and >= dim: the 16 groups of a wave then read 64
// different banks. grid: any (waves stride over the queries), block 1024, LDS route_dist_roots_lds_bytes(dim, n).
constexpr uint32_t kRootsWaves = 16;
inline uint32_t route_dist_roots_stride(uint32_t dim) { return dim + ((68 - (dim & 63)) & 63); }
inline size_t route_dist_roots_lds_bytes(uint32_t dim, uint32_t n) {
    return ((size_t)n * route_dist_roots_stride(dim) + (size_t)kRootsWaves * ((dim + 3) & ~3u)) * sizeof(float);
}
__global__ void __launch_bounds__(1024)
route_dist_roots_kernel(IndexDev ix, const float *queries, uint32_t nq, const uint32_t *shared_fr, uint32_t n, uint32_t cstride,
                        float *fdist, uint32_t fstride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t D = ix.dim, D4 = (D + 3) & ~3u;
    float *cs = reinterpret_cast<float *>(smem);
    float *qw = cs + (size_t)n * cstride + (threadIdx.x >> 6) * D4; // this wave's query
#pragma unroll 4
    for (uint32_t row = threadIdx.x >> 6; row < n; row += kRootsWaves) { // a wave per row: coalesced
        const float *c = ix.centroids + (uint64_t)shared_fr[row] * D;
        for (uint32_t i = threadIdx.x & 63; i < D; i += 64) cs[row * cstride + i] = c[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, sub = lane & 3, g = lane >> 2, n4 = n & ~3u;
    const bool vec = (D & 3) == 0;
    for (uint32_t q = blockIdx.x * kRootsWaves + (threadIdx.x >> 6); q < nq; q += gridDim.x * kRootsWaves) { // wave-uniform
        __builtin_amdgcn_wave_barrier(); // the last query's reads are behind us (LDS serves a wave in order)
        for (uint32_t i = lane; i < D; i += 64) qw[i] = queries[(uint64_t)q * D + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j0 = 0; j0 < n; j0 += 16) {
            const uint32_t j = j0 + g;
            const bool act = j < n;
            const float *c = cs + (act ? j : 0) * cstride;
            float d;
            if (j0 + 16 <= n4) // (wave-uniform) sixteen nodes of the batch-of-four arithmetic
                d = l2_batch4_group4(qw, c, D, sub);
            else if (j0 >= n4) // the 1..3 nodes past the last whole batch: l2_squared
                d = neon16_group4<true>(qw, c, D, sub, vec);
            else {
                const float db = l2_batch4_group4(qw, c, D, sub), ds = neon16_group4<true>(qw, c, D, sub, vec);
                d = j < n4 ? db : ds;
            }
            if (act && sub == 0) fdist[(uint64_t)q * fstride + j] = d;
        }
    }
}

// What route_select_reg_kernel needs to evaluate a query's exact distances itself before it selects (the two kernels
// that serve queries the matrix-core filter could not decide, as one launch): null queries = distances are in fdist.
// seg != null: such a query may so far only have its segment table (small[q] == 0): its frontier is written out first
// (route_materialise_kernel's work), into `frontier`.
struct FusedDist {
    const float *queries = nullptr;
    const float *qnorm = nullptr;
    float *fdist = nullptr;
    const uint32_t *seg = nullptr;
    const uint32_t *small = nullptr;
    uint32_t *frontier = nullptr;
    uint32_t pstride = 0;
};

inline size_t route_select_reg_lds_bytes(uint32_t scap, uint32_t fused_dim = 0) { // route_select_reg_kernel: four radix histograms (+ the query)
    return (size_t)scap * 8 + 4 * 256 * 4 + 32 * 4 + 8 * 4 + 4 * ((size_t)scap + 1) * 4 + 16 + ((size_t)fused_dim + 4) * 4;
}

inline size_t route_select_lds_bytes(uint32_t scap) {
    return (size_t)scap * 8 + 256 * 4 + 32 * 4 + 8 * 4 + 4 * ((size_t)scap + 1) * 4 + 16;
}

// Per query: choose the nodes to descend into (hierarchical.rs:393-408) and expand them.
//   n <= probes (and not stats_variant): all nodes, frontier order.
//   otherwise: the min(n, probes) smallest by (distance, frontier position), in that order.
// scap (a power of two <= kSortCap) bounds the sorted set: min(n, probes) <= scap whenever n > scap
// is possible. One block per query; dynamic LDS = route_select_lds_bytes(scap).
// shared_fr != null: every query's frontier is shared_fr[0..shared_n). next_frontier == null: the
// children are not written (see route_expand).
__global__ void __launch_bounds__(256)
route_select_kernel(IndexDev ix, const uint32_t *frontier, uint32_t fstride, const uint32_t *fcount,
                    const uint32_t *shared_fr, uint32_t shared_n, const float *fdist, uint32_t probes,
                    uint32_t stats_variant, uint32_t scap, uint32_t *next_frontier, uint32_t *next_fcount,
                    uint32_t *leaves, uint32_t lstride, uint32_t *lcount, const uint32_t *only_flag, PairOut po) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);                 // [scap]
    uint32_t *hist = reinterpret_cast<uint32_t *>(skeys + scap);         // [256]
    uint32_t *wtmp = hist + 256;                                         // [2 * waves]
    uint32_t *misc = wtmp + 32;                                          // [8]
    uint32_t *xs = misc + 8;                                             // [4 * (scap + 1)]
    const uint32_t q = blockIdx.x;
    if (only_flag && !only_flag[q]) return;
    const uint32_t n = shared_fr ? shared_n : fcount[q];
    const uint32_t tid = threadIdx.x;
    if (n == 0) {
        if (tid == 0 && next_frontier) next_fcount[q] = 0;
        if (po.info) {
            for (uint32_t i = tid; i < po.pstride; i += blockDim.x) po.info[(uint64_t)q * po.pstride + i] = 0xFFFFFFFFu;
            if (tid == 0) po.small[q] = 1;
        }
        return;
    }
    const uint32_t *fr = shared_fr ? shared_fr : frontier + (uint64_t)q * fstride;
    const float *fd = fdist + (uint64_t)q * fstride;
    uint32_t nsel;
    const bool keep_all = !stats_variant && n <= probes;
    if (keep_all) {
        nsel = n; // n <= this level's bound <= scap
        for (uint32_t i = tid; i < n; i += blockDim.x) skeys[i] = i;
        __syncthreads();
    } else {
        nsel = min(n, probes);
        uint32_t npad;
        if (n <= scap) {
            npad = next_pow2(n);
            for (uint32_t i = tid; i < npad; i += blockDim.x)
                skeys[i] = i < n ? (((uint64_t)f32_key(fd[i]) << 32) | i) : ~0ull;
            __syncthreads();
        } else {
            uint32_t budget;
            const uint32_t T = block_radix_select(n, nsel, [&](uint32_t i) { return f32_key(fd[i]); }, hist, misc, budget);
            ordered_threshold_compact(
                n, T, budget, [&](uint32_t i) { return f32_key(fd[i]); },
                [&](uint32_t i, uint32_t kv, uint32_t rank) { skeys[rank] = ((uint64_t)kv << 32) | i; }, wtmp);
            npad = next_pow2(nsel);
            for (uint32_t i = nsel + tid; i < npad; i += blockDim.x) skeys[i] = ~0ull;
            __syncthreads();
        }
        block_sort_u64(skeys, npad);
    }
    route_expand(ix, fr, skeys, nsel, xs, scap + 1,
                 next_frontier ? next_frontier + (uint64_t)q * fstride : nullptr, fstride, next_fcount + q,
                 leaves + (uint64_t)q * lstride, lstride, lcount + q, po, q);
}


// route_select_kernel for levels whose frontiers hold at most KPT * THREADS nodes: the keys
// are read once and kept in registers, radix passes whose digit is the same for every key are
// skipped (no counting, no same-address atomics), and the winners are collected in any order
// before the sort by (distance, frontier position) — unless the keys equal to the threshold
// outnumber the places left for them, in which case the earliest positions must win and the
// ordered compaction decides. Same arguments, same result; dynamic LDS = route_select_lds_bytes(scap).
template <int THREADS, int KPT>
__global__ void __launch_bounds__(THREADS)
route_select_reg_kernel(IndexDev ix, const uint32_t *frontier, uint32_t fstride, const uint32_t *fcount,
                        const uint32_t *shared_fr, uint32_t shared_n, const float *fdist, uint32_t probes,
                        uint32_t stats_variant, uint32_t scap, uint32_t *next_frontier, uint32_t *next_fcount,
                        uint32_t *leaves, uint32_t lstride, uint32_t *lcount, const uint32_t *only_flag, PairOut po,
                        FusedDist fused = FusedDist()) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);                 // [scap]
    uint32_t *hist = reinterpret_cast<uint32_t *>(skeys + scap);         // [4 * 256]
    uint32_t *wtmp = hist + 4 * 256;                                     // [2 * waves]
    uint32_t *misc = wtmp + 32;                                          // [8]
    uint32_t *xs = misc + 8;                                             // [4 * (scap + 1)]
    const uint32_t q = blockIdx.x;
    if (only_flag && !only_flag[q]) return;
    const uint32_t n = shared_fr ? shared_n : fcount[q];
    const uint32_t tid = threadIdx.x;
    if (fused.seg && !fused.small[q]) { // (the selection's key and counter arrays serve as the table's LDS until then)
        uint32_t *st = reinterpret_cast<uint32_t *>(smem);
        const uint32_t *sg = fused.seg + (uint64_t)q * seg_stride(fused.pstride);
        for (uint32_t i = tid; i < seg_stride(fused.pstride); i += blockDim.x) st[i] = sg[i];
        __syncthreads();
        const uint32_t nf = min(n, fstride);
        for (uint32_t o = tid; o < nf; o += blockDim.x) fused.frontier[(uint64_t)q * fstride + o] = frontier_node(ix, st, fused.pstride, o);
        __threadfence_block();
        __syncthreads();
    }
    if (fused.queries && (stats_variant || n > probes)) {
        // the query's exact distances first (route_dist_kernel's work for one block per query), then the selection reads
        // them back: written and read by this block, a barrier apart
        float *qs = reinterpret_cast<float *>(reinterpret_cast<uintptr_t>(xs + 4 * ((size_t)scap + 1) + 3) & ~(uintptr_t)15);
        route_dist_block(ix, qs, fused.queries + (uint64_t)q * ix.dim, shared_fr ? shared_fr : frontier + (uint64_t)q * fstride,
                         fused.fdist + (uint64_t)q * fstride, n, probes, stats_variant,
                         ix.metric == kMetricCosine ? fused.qnorm[q] : 0.f, 0, 1);
        __threadfence_block();
        __syncthreads();
    }
    if (n == 0) {
        if (tid == 0 && next_frontier) next_fcount[q] = 0;
        if (po.info) {
            for (uint32_t i = tid; i < po.pstride; i += blockDim.x) po.info[(uint64_t)q * po.pstride + i] = 0xFFFFFFFFu;
            if (tid == 0) po.small[q] = 1;
        }
        return;
    }
    const uint32_t *fr = shared_fr ? shared_fr : frontier + (uint64_t)q * fstride;
    const float *fd = fdist + (uint64_t)q * fstride;
    CUDDLE_STAMP(3, 0);
    uint32_t nsel;
    const bool keep_all = !stats_variant && n <= probes;
    if (keep_all) {
        nsel = n;
        for (uint32_t i = tid; i < n; i += blockDim.x) skeys[i] = i;
        __syncthreads();
    } else {
        nsel = min(n, probes);
        uint32_t npad;
        if (n <= scap) {
            npad = next_pow2(n);
            for (uint32_t i = tid; i < npad; i += blockDim.x)
                skeys[i] = i < n ? (((uint64_t)f32_key(fd[i]) << 32) | i) : ~0ull;
            __syncthreads();
        } else {
            uint32_t keys[KPT];
            float kv[KPT]; // all loads first (see route_prefilter_reg_kernel); n > scap >= 1 here
#pragma unroll
            for (int t = 0; t < KPT; ++t) kv[t] = fd[min((uint32_t)t * THREADS + tid, n - 1)];
#pragma unroll
            for (int t = 0; t < KPT; ++t) keys[t] = (uint32_t)t * THREADS + tid < n ? f32_key(kv[t]) : 0xFFFFFFFFu;
            if (tid < 2) misc[2 + tid] = 0; // winner counter, tie counter (visible after the barriers inside the select)
            CUDDLE_STAMP(3, 1);
            uint32_t want;
            const uint32_t prefix = reg_radix_select<THREADS, KPT>(keys, n, nsel, hist, wtmp, want);
            CUDDLE_STAMP(3, 2);
            const uint32_t T = prefix, budget = want;
            uint32_t my_ties = 0;
#pragma unroll
            for (int t = 0; t < KPT; ++t) my_ties += (t * THREADS + tid < n && keys[t] == T) ? 1u : 0u;
            if (my_ties) atomicAdd(&misc[3], my_ties);
            __syncthreads();
            if (misc[3] == budget) { // every key equal to the threshold is selected: order is the sort's business
#pragma unroll
                for (int t = 0; t < KPT; ++t) {
                    const uint32_t i = t * THREADS + tid;
                    if (i < n && keys[t] <= T) skeys[atomicAdd(&misc[2], 1u)] = ((uint64_t)keys[t] << 32) | i;
                }
            } else {
                ordered_threshold_compact(
                    n, T, budget, [&](uint32_t i) { return f32_key(fd[i]); },
                    [&](uint32_t i, uint32_t kv, uint32_t rank) { skeys[rank] = ((uint64_t)kv << 32) | i; }, wtmp);
            }
            npad = next_pow2(nsel);
            for (uint32_t i = nsel + tid; i < npad; i += blockDim.x) skeys[i] = ~0ull;
            __syncthreads();
        }
        CUDDLE_STAMP(3, 3);
        block_sort_u64(skeys, npad);
    }
    CUDDLE_STAMP(3, 4);
    route_expand(ix, fr, skeys, nsel, xs, scap + 1,
                 next_frontier ? next_frontier + (uint64_t)q * fstride : nullptr, fstride, next_fcount + q,
                 leaves + (uint64_t)q * lstride, lstride, lcount + q, po, q);
    CUDDLE_STAMP(3, 5);
}

// ---------------------------------------------------------------------------------------------
// grouped exact distances for a level whose frontiers differ per query
// ---------------------------------------------------------------------------------------------
// The frontier of such a level is, per query, the children of the nodes chosen one level up. Many
// queries choose the same parents, so the (query, parent) pairs are grouped by parent and one block
// evaluates kGroupNQ queries of a group against the parent's children: every centroid row is read
// once per block instead of once per query. The arithmetic per (query, child) is distance(), as
// search() uses for frontiers of more than 100 nodes (hierarchical.rs:355-365); queries with
// smaller frontiers are not grouped (PairOut::small).

// Exclusive prefix sums over the per-parent pair counts: poff = pairs, ioff = work items (blocks of
// up to kGroupPairs pairs of one parent); items[item] = (parent, first pair, pairs, 0), or, with
// child_rows (per parent: table row of its first child, number of children) given, (first pair,
// pairs, row, children) for route_gemm_grouped_kernel. The per-parent arrays hold n + 1 entries.
// Three launches, so that a level of millions of parents is scanned by as many blocks as it needs:
//   pair_block_sums_kernel  grid nb = ceil(n / 1024), block 256: bsum[b], bsum[nb + b] = pairs, items of block b's parents
//   pair_block_scan_kernel  one block of 1024: bsum -> exclusive offsets in place; totals to poff[n], ioff[n]
//   pair_scan_kernel        grid nb, block 256: offsets of each parent, and its items
constexpr uint32_t kPairScanChunk = 1024; // parents per block: 4 per thread
// Counters per parent (as a shift) for a level of `width` parents: up to 64, while the level stays under 8192 counters.
__host__ __device__ inline uint32_t pair_shift_of(uint32_t width) {
    uint32_t sh = 0;
    while (sh < 6 && ((uint64_t)width << (sh + 1)) <= 8192) ++sh;
    return sh;
}

__device__ __forceinline__ uint32_t pair_items_of(uint32_t c) { return (c + kGroupPairs - 1) / kGroupPairs; }

// Block-wide exclusive scan of (a, b) over 256 threads; returns the block totals in (ta, tb).
__device__ __forceinline__ void block256_exclusive_scan2(uint32_t &a, uint32_t &b, uint32_t &ta, uint32_t &tb, uint32_t *lds8) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t ia = wave_inclusive_sum(a), ib = wave_inclusive_sum(b);
    if (lane == 63) {
        lds8[w] = ia;
        lds8[4 + w] = ib;
    }
    __syncthreads();
    uint32_t oa = 0, ob = 0;
    ta = tb = 0;
    for (uint32_t v = 0; v < 4; ++v) {
        if (v < w) {
            oa += lds8[v];
            ob += lds8[4 + v];
        }
        ta += lds8[v];
        tb += lds8[4 + v];
    }
    a = oa + ia - a;
    b = ob + ib - b;
}

__global__ void __launch_bounds__(256)
pair_block_sums_kernel(const uint32_t *cnt, uint32_t n, uint32_t *bsum, uint32_t nb) {
    __shared__ uint32_t lds8[8];
    const uint32_t base = blockIdx.x * kPairScanChunk + threadIdx.x * 4;
    uint32_t ap = 0, ai = 0;
    for (uint32_t k = 0; k < 4; ++k)
        if (base + k < n) {
            const uint32_t c = cnt[base + k];
            ap += c;
            ai += pair_items_of(c);
        }
    uint32_t tp, ti;
    block256_exclusive_scan2(ap, ai, tp, ti, lds8);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = tp;
        bsum[nb + blockIdx.x] = ti;
    }
}

__global__ void __launch_bounds__(1024)
pair_block_scan_kernel(uint32_t *bsum, uint32_t nb, uint32_t n, uint32_t *poff, uint32_t *ioff) {
    __shared__ uint32_t sp[1024], si[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t carry_p = 0, carry_i = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) { // block-uniform trip count
        const uint32_t b = b0 + tid;
        const uint32_t vp = b < nb ? bsum[b] : 0, vi = b < nb ? bsum[nb + b] : 0;
        sp[tid] = vp;
        si[tid] = vi;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t xp = tid >= o ? sp[tid - o] : 0, xi = tid >= o ? si[tid - o] : 0;
            __syncthreads();
            sp[tid] += xp;
            si[tid] += xi;
            __syncthreads();
        }
        if (b < nb) {
            bsum[b] = carry_p + sp[tid] - vp;
            bsum[nb + b] = carry_i + si[tid] - vi;
        }
        carry_p += sp[1023];
        carry_i += si[1023];
        __syncthreads();
    }
    if (tid == 0) {
        poff[n] = carry_p;
        ioff[n] = carry_i;
    }
}

__global__ void __launch_bounds__(256)
pair_scan_kernel(const uint32_t *cnt, uint32_t n, const uint32_t *bsum, uint32_t nb, uint32_t *poff, uint32_t *ioff,
                 uint4 *items, const uint2 *child_rows, uint32_t *work_bins = nullptr, const uint32_t *vec_off = nullptr,
                 uint32_t shift = 0) { // shift: entry i counts for parent i >> shift (PairOut::shift)
    __shared__ uint32_t lds8[8];
    static_assert(kWorkBins == 256, "one work class per thread of the block");
    __shared__ uint32_t wb[kWorkBins]; // this block's items per work class (one global add per class and block)
    if (work_bins) wb[threadIdx.x] = 0; // (256 threads, 256 classes; visible after the scan's barrier)
    const uint32_t base = blockIdx.x * kPairScanChunk + threadIdx.x * 4;
    uint32_t c4[4], ap = 0, ai = 0;
    for (uint32_t k = 0; k < 4; ++k) {
        c4[k] = base + k < n ? cnt[base + k] : 0;
        ap += c4[k];
        ai += pair_items_of(c4[k]);
    }
    uint32_t tp, ti;
    block256_exclusive_scan2(ap, ai, tp, ti, lds8);
    uint32_t carry_p, carry_i;
    if (bsum) {
        carry_p = bsum[blockIdx.x];
        carry_i = bsum[nb + blockIdx.x];
    } else {
        // bsum == null (a level of few chunks): the block sums the chunks before its own itself, which costs less than
        // the two launches that would hand it the sums; the last block also writes the totals
        __shared__ uint32_t red[8];
        const uint32_t lim = blockIdx.x * kPairScanChunk;
        uint32_t sp = 0, si = 0;
        for (uint32_t i0 = threadIdx.x; i0 < lim; i0 += 256 * 4) {
            uint32_t c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = cnt[min(i0 + 256 * u, lim - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 256 * u < lim) {
                    sp += c[u];
                    si += pair_items_of(c[u]);
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sp += __shfl_xor(sp, o, 64);
            si += __shfl_xor(si, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6] = sp;
            red[4 + (threadIdx.x >> 6)] = si;
        }
        __syncthreads();
        carry_p = red[0] + red[1] + red[2] + red[3];
        carry_i = red[4] + red[5] + red[6] + red[7];
        if (blockIdx.x == nb - 1 && threadIdx.x == 0) {
            poff[n] = carry_p + tp;
            ioff[n] = carry_i + ti;
        }
    }
    uint32_t rp = carry_p + ap, ri = carry_i + ai;
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = base + k;
        if (i >= n) break;
        const uint32_t c = c4[k], ni = pair_items_of(c);
        poff[i] = rp;
        ioff[i] = ri;
        if (ni) {
            const uint2 cr = child_rows ? child_rows[i >> shift] : make_uint2(0u, 0u);
            // work_bins != null (parents are leaves, vec_off their vector ranges): the items are also counted by work,
            // for item_scatter_kernel to put them in heaviest-first order
            const uint32_t nv = work_bins ? vec_off[i + 1] - vec_off[i] : 0u;
            for (uint32_t t = 0; t < ni; ++t) {
                const uint32_t first = rp + t * kGroupPairs, np = min((uint32_t)kGroupPairs, c - t * kGroupPairs);
                items[ri + t] = child_rows ? make_uint4(first, np, cr.x, cr.y) : make_uint4(i >> shift, first, np, 0u);
                if (work_bins) atomicAdd(&wb[item_work_bin(np, nv)], 1u);
            }
        }
        rp += c;
        ri += ni;
    }
    if (work_bins) {
        __syncthreads();
        if (wb[threadIdx.x]) atomicAdd(&work_bins[threadIdx.x], wb[threadIdx.x]);
    }
}

// Scatter the pairs into parent order (order inside a group is immaterial).
__global__ void pair_fill_kernel(const uint32_t *info, const uint32_t *parent, const uint32_t *slot, uint32_t total,
                                 const uint32_t *poff, uint32_t *sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t v = info[i];
    if (v == 0xFFFFFFFFu) return;
    sorted[poff[parent[i]] + slot[i]] = v;
}

// One block per work item = (parent, up to kGroupPairs of its pairs); blocks past the last item
// exit. block = 256: one 4-lane group per child (64 children per pass).
//   NBLK > 0: dim == 16 * NBLK. The group keeps its slice of the child's centroid in registers
//     while the item's pairs go by in tiles of kGroupNQ queries staged in LDS; the next tile's
//     queries are fetched into registers while the current tile is evaluated (two LDS buffers, one
//     barrier per tile), four queries at a time to bound the accumulator registers.
//   NBLK == 0: any dim. The slice is re-read in segments of 8 chunks for every tile.
// METRIC is the index metric. dynamic LDS = 2 * kGroupNQ * pad4(dim) floats.
template <int NBLK, uint32_t METRIC>
__global__ void __launch_bounds__(256, (NBLK > 8 ? 2 : 4) - (METRIC == kMetricCosine ? 1 : 0)) // waves per SIMD the registers must allow
route_dist_grouped_kernel(IndexDev ix, const float *queries, const uint32_t *level_nodes, uint32_t n_parents,
                          const uint32_t *ioff, const uint4 *items, const uint32_t *sorted, float *fdist,
                          uint32_t fstride, const float *qnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qbuf = reinterpret_cast<float *>(smem);
    __shared__ uint32_t pk[kGroupPairs];
    constexpr bool L2 = METRIC == kMetricL2;
    const uint32_t item = blockIdx.x;
    const uint32_t n_items = ioff[n_parents];
    const uint4 rec = items[min(item, max(n_items, 1u) - 1)];
    if (item >= n_items) return;
    CUDDLE_STAMP(2, 0);
    const uint32_t p = rec.x, pb = rec.y, np = rec.z;
    const uint32_t D = NBLK > 0 ? 16u * NBLK : ix.dim, Dp = (D + 3) & ~3u, tid = threadIdx.x;
    if (tid < kGroupPairs) pk[tid] = sorted[pb + min(tid, np - 1)]; // unused slots repeat the last pair
    const uint32_t node = level_nodes[p];
    const uint32_t cb = ix.child_off[node], cc = ix.child_off[node + 1] - cb;
    const uint32_t sub = tid & 3, nblk = D >> 4;
    const bool vec = NBLK > 0 || (D & 3) == 0;
    constexpr int PF = NBLK > 0 ? (NBLK + 7) / 8 : 1; // float4 per thread and tile: kGroupNQ * 4 * NBLK / 256
    constexpr uint32_t d4 = NBLK > 0 ? 4u * NBLK : 1u; // float4 per query
    for (uint32_t c0 = 0; c0 < cc; c0 += 64) { // block-uniform trip counts throughout
        const uint32_t c = c0 + (tid >> 2);
        const bool act = c < cc;
        const float *row = ix.centroids + (uint64_t)ix.children[cb + (act ? c : 0)] * D;
        float4 y[NBLK > 0 ? NBLK : 1];
        float nb = 0.f;
        if (NBLK > 0) {
#pragma unroll
            for (int t = 0; t < NBLK; ++t) y[t] = *reinterpret_cast<const float4 *>(row + 4 * sub + 16 * t);
        }
        if (METRIC == kMetricCosine) nb = sqrtf(neon16_group4<false>(row, row, D, sub, vec));
        CUDDLE_STAMP(2, 1);
        __syncthreads(); // pk ready (first pass) / last tile of the previous pass consumed
        float4 pf[PF];
        // queries of tile t0 -> registers: float4 number f of the tile = tid + 256 k
        auto fetch = [&](uint32_t t0) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const uint32_t f = tid + 256 * k, j = f / d4, e = f - j * d4;
                pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < kGroupNQ)
                    pf[k] = *reinterpret_cast<const float4 *>(queries + (uint64_t)(pk[min(t0 + j, np - 1)] >> kPairOffBits) * D + 4 * e);
            }
        };
        auto stash = [&](float *qs) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const uint32_t f = tid + 256 * k, j = f / d4, e = f - j * d4;
                if (j < kGroupNQ) *reinterpret_cast<float4 *>(qs + j * Dp + 4 * e) = pf[k];
            }
        };
        if (NBLK > 0) {
            fetch(0);
            stash(qbuf);
            __syncthreads();
        }
        CUDDLE_STAMP(2, 2);
        for (uint32_t t0 = 0, it = 0; t0 < np; t0 += kGroupNQ, ++it) {
            float *qs = qbuf + (NBLK > 0 ? (it & 1) * kGroupNQ * Dp : 0);
            if (NBLK > 0) {
                if (t0 + kGroupNQ < np) fetch(t0 + kGroupNQ);
            } else {
                if (it) __syncthreads(); // previous tile consumed
                for (uint32_t j = 0; j < kGroupNQ; ++j) {
                    const float *qg = queries + (uint64_t)(pk[min(t0 + j, np - 1)] >> kPairOffBits) * D;
                    for (uint32_t i = tid; i < D; i += blockDim.x) qs[j * Dp + i] = qg[i];
                }
                __syncthreads();
            }
            float dres[kGroupNQ];
#pragma unroll
            for (uint32_t jh = 0; jh < kGroupNQ; jh += 4) { // four queries at a time
                __builtin_amdgcn_sched_barrier(0);
                // two-float vector arithmetic = the packed f32 instructions; each component is the
                // same IEEE operation as the scalar form
                v2f acc_lo[4], acc_hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc_lo[j] = acc_hi[j] = v2f{0.f, 0.f};
                const float *qh = qs + jh * Dp + 4 * sub;
                auto load_x = [&](float4 *x, uint32_t t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = ld4(qh + (uint32_t)j * Dp + 16 * t, vec);
                };
                auto step = [&](const float4 &yy, const float4 *x) {
                    const v2f ylo{yy.x, yy.y}, yhi{yy.z, yy.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v2f xlo{x[j].x, x[j].y}, xhi{x[j].z, x[j].w};
                        if (L2) {
                            const v2f dlo = xlo - ylo, dhi = xhi - yhi;
                            acc_lo[j] = __builtin_elementwise_fma(dlo, dlo, acc_lo[j]);
                            acc_hi[j] = __builtin_elementwise_fma(dhi, dhi, acc_hi[j]);
                        } else {
                            acc_lo[j] = __builtin_elementwise_fma(xlo, ylo, acc_lo[j]);
                            acc_hi[j] = __builtin_elementwise_fma(xhi, yhi, acc_hi[j]);
                        }
                    }
                };
                if (NBLK > 0) {
                    // the queries of chunk t + 1 are read from LDS while chunk t is evaluated; the
                    // scheduling barrier keeps the compiler from hoisting every read to the top
                    float4 xc[4], xn[4];
                    load_x(xc, 0);
#pragma unroll
                    for (int t = 0; t < NBLK; ++t) {
                        if (t + 1 < NBLK) load_x(xn, t + 1);
                        step(y[t], xc);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) xc[j] = xn[j];
                    }
                } else {
                    for (uint32_t s0 = 0; s0 < nblk; s0 += 8) { // 8 row loads in flight per lane
                        float4 ys[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) ys[t] = s0 + t < nblk ? ld4(row + 4 * sub + 16 * (s0 + t), vec) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (s0 + t < nblk) {
                                float4 x[4];
                                load_x(x, s0 + t);
                                step(ys[t], x);
                            }
                    }
                }
                float4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = make_float4(acc_lo[j].x, acc_lo[j].y, acc_hi[j].x, acc_hi[j].y);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = group4_reduce(acc[j]);
                    const float *qa = qs + (jh + j) * Dp;
                    float d;
                    if (L2) {
                        d = NBLK > 0 ? r : neon_tail<true>(r, qa, row, nblk << 4, D);
                    } else {
                        const float dot = NBLK > 0 ? r : neon_tail<false>(r, qa, row, nblk << 4, D);
                        if (METRIC == kMetricDot) {
                            d = -dot;
                        } else {
                            const float na = qnorm[pk[min(t0 + jh + j, np - 1)] >> kPairOffBits]; // sqrtf(dot(q, q)), once per query
                            d = 1.0f - (dot / (na * nb));
                        }
                    }
                    dres[jh + j] = d;
                }
            }
            // the next tile's queries go to LDS before this tile's results are stored: waiting for
            // the fetch must not also wait for the stores to be acknowledged
            if (NBLK > 0 && t0 + kGroupNQ < np) {
                stash(qbuf + ((it + 1) & 1) * kGroupNQ * Dp); // that buffer's tile was consumed before the last barrier
                __syncthreads();
            }
            if (act && sub == 0) {
#pragma unroll
                for (int j = 0; j < kGroupNQ; ++j) {
                    const uint32_t e = pk[min(t0 + j, np - 1)], off = (e & ((1u << kPairOffBits) - 1)) + c;
                    if (t0 + j < np && off < fstride) fdist[(uint64_t)(e >> kPairOffBits) * fstride + off] = dres[j];
                }
            }
        }
    }
    CUDDLE_STAMP(2, 3);
}

// ---------------------------------------------------------------------------------------------
// MFMA path
// ---------------------------------------------------------------------------------------------

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmBM = 128;

// The contraction runs on the bf16 matrix cores (16 times the rate of the f32 ones) at close to f32
// accuracy: an f32 value x is carried as two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi), both
// rounded to nearest even, so |x - hi - lo| <= 2^-16 |x|. Of the four partial products of x y the
// kernel sums hi*hi on one accumulator and hi*lo + lo*hi on a second one (bf16 products are exact in
// f32); what it drops (lo*lo and the two residuals) is at most 3.0001 * 2^-16 |x y| per product.
// Split operand layout, per row: K' = K rounded up to 64 values, stored as K'/8 groups of 32 bytes
// = the 8 hi halves followed by the 8 lo halves of the same 8 values; zeros beyond K.

inline uint32_t split_kp(uint32_t dim) { return (dim + 63) / 64 * 64; } // K' of a split row

__device__ __forceinline__ uint32_t bf16_round(float x) {
    const uint32_t b = __float_as_uint(x);
    return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void bf16_split_store(uint16_t *row, uint32_t i, float v) {
    const uint32_t hi = bf16_round(v);
    const uint32_t lo = bf16_round(v - __uint_as_float(hi << 16));
    uint16_t *g = row + (i >> 3) * 16 + (i & 7);
    g[0] = (uint16_t)hi;
    g[8] = (uint16_t)lo;
}

// Per-level table for the contraction (built once per index), in split form: row j = centroid of
// the j-th node of the level (breadth-first order = the frontier order every query sees when all
// earlier levels keep all their nodes) minus mu (L2: the level's mean centroid, which keeps the norms
// and so the rounding error of the norm expansion small; other metrics: mu = 0), plus the squared
// norm of the f32 row. One wave per row; grid = ceil(n / 4), block 256. level_nodes == null: row j of
// `centroids` itself.
__global__ void __launch_bounds__(256)
level_table_kernel(const float *centroids, const uint32_t *level_nodes, uint32_t n, uint32_t dim, uint32_t kp,
                   const float *mu, uint16_t *table, float *sqn) {
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const float *c = centroids + (uint64_t)(level_nodes ? level_nodes[j] : j) * dim;
    uint16_t *t = table + (uint64_t)j * kp * 2;
    float s = 0.f;
    for (uint32_t i = lane; i < kp; i += 64) {
        const float v = i < dim ? c[i] - mu[i] : 0.f;
        bf16_split_store(t, i, v);
        s = __builtin_fmaf(v, v, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sqn[j] = s;
}

// q' = q - mu in split form and |q'|^2 for every query of the batch (one wave per query).
__global__ void __launch_bounds__(256)
center_queries_kernel(const float *queries, const float *mu, uint32_t nq, uint32_t dim, uint32_t kp, uint16_t *out,
                            float *sqn) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    uint16_t *row = out + (uint64_t)q * kp * 2;
    float s = 0.f;
    if ((dim & 3) == 0 && (kp & 7) == 0) {
        // eight values per lane and step: two 16-byte loads, one 16-byte store each for the hi and the lo unit
        const float *src = queries + (uint64_t)q * dim;
        for (uint32_t g = lane; g < kp / 8; g += 64) {
            float v[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = 8 * g + 4 * h;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f), m = x;
                if (i < dim) { // dim is a multiple of 4: the four values are there or not
                    x = *reinterpret_cast<const float4 *>(src + i);
                    m = *reinterpret_cast<const float4 *>(mu + i);
                }
                v[4 * h] = x.x - m.x; v[4 * h + 1] = x.y - m.y; v[4 * h + 2] = x.z - m.z; v[4 * h + 3] = x.w - m.w;
            }
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t h0 = bf16_round(v[2 * c]), h1 = bf16_round(v[2 * c + 1]);
                hi[c] = h0 | (h1 << 16);
                lo[c] = bf16_round(v[2 * c] - __uint_as_float(h0 << 16)) | (bf16_round(v[2 * c + 1] - __uint_as_float(h1 << 16)) << 16);
                s = __builtin_fmaf(v[2 * c], v[2 * c], s);
                s = __builtin_fmaf(v[2 * c + 1], v[2 * c + 1], s);
            }
            uint4 *dst = reinterpret_cast<uint4 *>(row + g * 16);
            dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            dst[1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    } else {
        for (uint32_t i = lane; i < kp; i += 64) {
            float v = i < dim ? queries[(uint64_t)q * dim + i] - mu[i] : 0.f;
            bf16_split_store(row, i, v);
            s = __builtin_fmaf(v, v, s);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sqn[q] = s;
}

// ---- single-pass half-precision operands (levels every query sees whole) -----------------------
// A level near the top of the tree is wide compared with probes, its nodes are spread out, and the
// gap between the probes-th and the (probes + margin)-th distance is large compared with what one
// rounding of each operand to 11 significant bits costs. There the filter needs one matrix-core
// product per value pair instead of the split form's three, and half the operand bytes.
// Every row (query or node) is scaled by its own power of two so that its largest component lands
// in [2^13, 2^14): then a component is below fp16's normal range only if it is 2^-27 of the row's
// largest, and whatever the conversion does with it (round to a subnormal: 2^-25, flush: 2^-14 in
// scaled units) is at most 2^-27 |row|_inf in the row's own units. Per product x y:
//   |x^ y^ - x y| <= (2^-10 + 2^-22) |x y| + 2^-27 (|x| |b|_inf + |y| |a|_inf) (1 + 2^-11),
// summed over a row pair: <= (2^-10 + 2^-22) sum|x y| + 2^-26 sqrt(D) |a| |b| (1 + 2^-11); fp16
// products are exact in f32. route_error_bound(half = 1) collects this as 1.01 * 2^-10.

// The power of two 2^se with m * 2^se in [2^13, 2^14) (m > 0 finite; exponent clamped so that the
// scale and its inverse are normal floats); m = 0 or not finite: 1.
__device__ __forceinline__ int half_scale_exp(float m) {
    if (!(m > 0.f) || !(m < __builtin_inff())) return 0;
    int e;
    (void)frexpf(m, &e); // m = f * 2^e, f in [0.5, 1)
    return max(-120, min(120, 14 - e));
}

// Row j of the level (see level_table_kernel) minus mu as fp16, scaled; inv[j] = the inverse scale.
// One wave per row; grid = ceil(n / 4), block 256. table: [n][kp] halfs (kp a multiple of 64, zero padded).
__global__ void __launch_bounds__(256)
level_table16_kernel(const float *centroids, const uint32_t *level_nodes, uint32_t n, uint32_t dim, uint32_t kp,
                     const float *mu, _Float16 *table, float *inv) {
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const float *c = centroids + (uint64_t)(level_nodes ? level_nodes[j] : j) * dim;
    float m = 0.f;
    for (uint32_t i = lane; i < dim; i += 64) m = fmaxf(m, fabsf(c[i] - mu[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int se = half_scale_exp(m);
    const float sc = ldexpf(1.0f, se);
    _Float16 *t = table + (uint64_t)j * kp;
    for (uint32_t i = lane; i < kp; i += 64) t[i] = (_Float16)(i < dim ? (c[i] - mu[i]) * sc : 0.f);
    if (lane == 0) inv[j] = ldexpf(1.0f, -se);
}

// q' = q - mu as scaled fp16, |q'|^2 (of the f32 values) and the inverse scale, one wave per query.
__global__ void __launch_bounds__(256)
center_queries16_kernel(const float *queries, const float *mu, uint32_t nq, uint32_t dim, uint32_t kp, _Float16 *out,
                        float *sqn, float *inv) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const float *src = queries + (uint64_t)q * dim;
    _Float16 *row = out + (uint64_t)q * kp;
    float s = 0.f, m = 0.f;
    const bool vec = (dim & 3) == 0 && (kp & 7) == 0;
    if (vec && kp <= 1024) {
        // rows of up to 1024 values: the lane's (at most four) 16-byte pieces of q and mu are all requested at once and
        // stay in registers for both passes - one trip to memory instead of one per piece and pass
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        float4 x[4], u[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = min(4 * lane + 256 * c, dim - 4);
            x[c] = *reinterpret_cast<const float4 *>(src + i);
            u[c] = *reinterpret_cast<const float4 *>(mu + i);
        }
        float v[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in = 4 * lane + 256 * c < dim;
            v[c][0] = in ? x[c].x - u[c].x : 0.f; v[c][1] = in ? x[c].y - u[c].y : 0.f;
            v[c][2] = in ? x[c].z - u[c].z : 0.f; v[c][3] = in ? x[c].w - u[c].w : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = __builtin_fmaf(v[c][e], v[c][e], s);
                m = fmaxf(m, fabsf(v[c][e]));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            m = fmaxf(m, __shfl_xor(m, o, 64));
        }
        const int se = half_scale_exp(m);
        const float sc = ldexpf(1.0f, se);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = 4 * lane + 256 * c;
            if (i < kp) {
                f16x4 o = {(_Float16)(v[c][0] * sc), (_Float16)(v[c][1] * sc), (_Float16)(v[c][2] * sc), (_Float16)(v[c][3] * sc)};
                *reinterpret_cast<f16x4 *>(row + i) = o;
            }
        }
        if (lane == 0) {
            sqn[q] = s;
            inv[q] = ldexpf(1.0f, -se);
        }
        return;
    }
    if (vec) {
        for (uint32_t i = 4 * lane; i < dim; i += 256) {
            const float4 x = *reinterpret_cast<const float4 *>(src + i), u = *reinterpret_cast<const float4 *>(mu + i);
            const float v0 = x.x - u.x, v1 = x.y - u.y, v2 = x.z - u.z, v3 = x.w - u.w;
            s = __builtin_fmaf(v0, v0, s); s = __builtin_fmaf(v1, v1, s); s = __builtin_fmaf(v2, v2, s); s = __builtin_fmaf(v3, v3, s);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
        }
    } else {
        for (uint32_t i = lane; i < dim; i += 64) {
            const float v = src[i] - mu[i];
            s = __builtin_fmaf(v, v, s);
            m = fmaxf(m, fabsf(v));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        m = fmaxf(m, __shfl_xor(m, o, 64));
    }
    const int se = half_scale_exp(m);
    const float sc = ldexpf(1.0f, se);
    if (vec) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        for (uint32_t i = 4 * lane; i < kp; i += 256) { // the same lane re-reads what it read above
            f16x4 o = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (i < dim) {
                const float4 x = *reinterpret_cast<const float4 *>(src + i), u = *reinterpret_cast<const float4 *>(mu + i);
                o[0] = (_Float16)((x.x - u.x) * sc); o[1] = (_Float16)((x.y - u.y) * sc);
                o[2] = (_Float16)((x.z - u.z) * sc); o[3] = (_Float16)((x.w - u.w) * sc);
            }
            *reinterpret_cast<f16x4 *>(row + i) = o;
        }
    } else {
        for (uint32_t i = lane; i < kp; i += 64) row[i] = (_Float16)(i < dim ? (src[i] - mu[i]) * sc : 0.f);
    }
    if (lane == 0) {
        sqn[q] = s;
        inv[q] = ldexpf(1.0f, -se);
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int BK>
constexpr size_t route_gemm_lds_bytes() { return (size_t)(kGemmBM + 64 * NT) * (BK / 4 + 1) * 16; }

// Approximate distances of every query to every node of one level:
//   S = A * B^T from the split operands (A = centred queries [M][K'], B = level table [N][K']):
//       S = sum_k (ah bh) + sum_k (ah bl + al bh)   (v_mfma_f32_32x32x16_bf16, two accumulators per tile)
//   L2:     d~ = (|a|^2 + |b|^2) - 2 S
//   DOT:    d~ = -S
//   COSINE: d~ = 1 - S / (|a| |b|)
// Block tile 128 x (64 NT), K step BK (32 or 64), 4 waves as 2 x 2, each wave 64 x (32 NT) = 2 x NT
// MFMA tiles of 32 x 32 (NT = 2 has the higher arithmetic intensity, NT = 1 twice the blocks).
// LDS image per operand: [row][BK/8 groups of (8 hi, 8 lo)] in 16-byte units with one unit of padding
// per row, so that the 32 rows a half-wave reads with ds_read_b128 fall into distinct bank groups. Which
// 8 values of a 16-value step go to which half-wave is the same for A and B, which is all a dot
// product needs. Global tiles are prefetched into registers one K step ahead.
// Blocks are numbered for the XCDs' private L2s (block id % 8 = XCD): the 8 XCDs form rg row
// groups x (8 / rg) panel groups; an XCD works on its row group's tiles of A (kept in its L2: the
// host picks rg so that they take at most half of it) against its panel group's column panels of
// B, the row tiles of one panel running together so that the panel is fetched once.
// out: [M][ostride]. grid = 8 * ceil(gy / rg) * ceil(gx / (8 / rg)).
// One block tile of the contraction: acc + cor = A[by tile] * B[bx tile]^T (see route_gemm_kernel for
// the layouts). smem: route_gemm_lds_bytes<NT, BK>() bytes; starts with a barrier, so the caller may
// have used the LDS for something else before. arow != null: the tile's 128 rows of A are the rows
// arow[0..128) (an LDS table; by and M are not used). All 256 threads call.
template <int NT, int BK>
__device__ __forceinline__ void split_gemm_tile(const u32x4 *A, const u32x4 *B, uint32_t M, uint32_t N, uint32_t kp,
                                                uint32_t by, uint32_t bx, unsigned char *smem, f32x16 (&acc)[2][NT],
                                                f32x16 (&cor)[2][NT], const uint32_t *arow = nullptr) {
    constexpr int BN = 64 * NT, CH = BK / 4, RS = CH + 1, RPP = 256 / CH, PA = kGemmBM / RPP, PB = BN / RPP;
    u32x4 *As = reinterpret_cast<u32x4 *>(smem); // [128][RS]
    u32x4 *Bs = As + kGemmBM * RS;               // [BN][RS]
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    const uint32_t lr = tid / CH, c = tid % CH, rowu = kp / 4; // 16-byte units per split row
    const u32x4 *ga[PA], *gb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i)
        ga[i] = A + (uint64_t)(arow ? arow[lr + RPP * i] : min(by * kGemmBM + lr + RPP * i, M - 1)) * rowu + c;
#pragma unroll
    for (int i = 0; i < PB; ++i) gb[i] = B + (uint64_t)min(bx * BN + lr + RPP * i, N - 1) * rowu + c;
    u32x4 pa[PA], pb[PB];
    auto load_tile = [&](uint32_t u0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) pa[i] = ga[i][u0];
#pragma unroll
        for (int i = 0; i < PB; ++i) pb[i] = gb[i][u0];
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) As[(lr + RPP * i) * RS + c] = pa[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) Bs[(lr + RPP * i) * RS + c] = pb[i];
    };
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = cor[mt][nt][e] = 0.f;
    load_tile(0);
    for (uint32_t u0 = 0; u0 < rowu; u0 += CH) {
        __syncthreads(); // previous step's LDS reads are done
        store_tile();
        __syncthreads();
        if (u0 + CH < rowu) load_tile(u0 + CH);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const uint32_t g = 2 * (2 * ks + h); // this half-wave's group of 8 values: hi unit, then lo unit
            bf16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 *p = As + (wr * 64 + t * 32 + r) * RS + g;
                ah[t] = __builtin_bit_cast(bf16x8, p[0]);
                al[t] = __builtin_bit_cast(bf16x8, p[1]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4 *p = Bs + (wc * 32 * NT + t * 32 + r) * RS + g;
                bh[t] = __builtin_bit_cast(bf16x8, p[0]);
                bl[t] = __builtin_bit_cast(bf16x8, p[1]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cor[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], cor[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cor[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], cor[mt][nt], 0, 0, 0);
        }
    }
}

// d~ of one (query row, node) from the contraction's S and the two squared norms.
__device__ __forceinline__ float approx_distance(uint32_t metric, float s, float as, float bs) {
    if (metric == kMetricL2) return __builtin_fmaf(-2.0f, s, as + bs);
    if (metric == kMetricDot) return -s;
    return 1.0f - s / (sqrtf(as) * sqrtf(bs));
}

template <int NT, int BK>
__global__ void __launch_bounds__(256, NT == 1 ? 3 : 2)
route_gemm_kernel(const u32x4 *A, const float *Asq, const u32x4 *B, const float *Bsq, uint32_t M, uint32_t N,
                  uint32_t kp, uint32_t metric, float *out, uint32_t ostride, uint32_t gx, uint32_t gy, uint32_t rg) {
    constexpr int BN = 64 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7, local = id >> 3, pgroups = 8 / rg, rows_per = (gy + rg - 1) / rg;
    const uint32_t by = (xcd % rg) * rows_per + local % rows_per, bx = (local / rows_per) * pgroups + xcd / rg;
    if (bx >= gx || by >= gy) return;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    f32x16 acc[2][NT], cor[2][NT];
    split_gemm_tile<NT, BK>(A, B, M, N, kp, by, bx, smem, acc, cor);
    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t gn = bx * BN + wc * 32 * NT + nt * 32 + r;
            if (gn >= N) continue;
            const float bs = Bsq[gn];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t gm = by * kGemmBM + wr * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (gm >= M) continue;
                out[(uint64_t)gm * ostride + gn] = approx_distance(metric, acc[mt][nt][e] + cor[mt][nt][e], Asq[gm], bs);
            }
        }
}

// The contraction from single-pass half-precision operands (level_table16_kernel /
// center_queries16_kernel): S = (sum_k a^ b^) * ainv * binv on v_mfma_f32_32x32x16_f16, one
// accumulator per tile, d~ from S and the f32 squared norms as above. Same block tile (128 x 64, 4
// waves as 2 x 2; NT = 2: 128 x 128), XCD-aware block order and output as route_gemm_kernel; half the operand bytes per
// K step, so the blocks are lighter (LDS: (128 + 64) rows x (BK / 8 + 1) 16-byte units) and more of
// them are in flight per CU, which is what a loop bound by the latency of its tile loads needs.
// kp a multiple of BK.
template <int NT, int BK>
constexpr size_t route_gemm16_lds_bytes() { return (size_t)(kGemmBM + 64 * NT) * (BK / 8 + 1) * 16; }

template <int NT, int BK>
__global__ void __launch_bounds__(256, NT == 1 ? 4 : 3)
route_gemm16_kernel(const u32x4 *A, const float *Asq, const float *Ainv, const u32x4 *B, const float *Bsq, const float *Binv,
                    uint32_t M, uint32_t N, uint32_t kp, uint32_t metric, float *out, uint32_t ostride, uint32_t gx, uint32_t gy,
                    uint32_t rg) {
    constexpr int BN = 64 * NT, CH = BK / 8, RS = CH + 1, RPP = 256 / CH, PA = kGemmBM / RPP, PB = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *As = reinterpret_cast<u32x4 *>(smem); // [128][RS]
    u32x4 *Bs = As + kGemmBM * RS;               // [BN][RS]
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7, local = id >> 3, pgroups = 8 / rg, rows_per = (gy + rg - 1) / rg;
    const uint32_t by = (xcd % rg) * rows_per + local % rows_per, bx = (local / rows_per) * pgroups + xcd / rg;
    if (bx >= gx || by >= gy) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    const uint32_t lr = tid / CH, c = tid % CH, rowu = kp / 8; // 16-byte units (8 halfs) per row
    const u32x4 *ga[PA], *gb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ga[i] = A + (uint64_t)min(by * kGemmBM + lr + RPP * i, M - 1) * rowu + c;
#pragma unroll
    for (int i = 0; i < PB; ++i) gb[i] = B + (uint64_t)min(bx * BN + lr + RPP * i, N - 1) * rowu + c;
    u32x4 pa[PA], pb[PB];
    auto load_tile = [&](uint32_t u0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) pa[i] = ga[i][u0];
#pragma unroll
        for (int i = 0; i < PB; ++i) pb[i] = gb[i][u0];
    };
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    load_tile(0);
    for (uint32_t u0 = 0; u0 < rowu; u0 += CH) {
        __syncthreads(); // previous step's LDS reads are done
#pragma unroll
        for (int i = 0; i < PA; ++i) As[(lr + RPP * i) * RS + c] = pa[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) Bs[(lr + RPP * i) * RS + c] = pb[i];
        __syncthreads();
        if (u0 + CH < rowu) load_tile(u0 + CH);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const uint32_t g = 2 * ks + h; // this half-wave's 8 of the step's 16 values
            f16x8 a[2], b[NT];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = __builtin_bit_cast(f16x8, As[(wr * 64 + t * 32 + r) * RS + g]);
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = __builtin_bit_cast(f16x8, Bs[(wc * 32 * NT + t * 32 + r) * RS + g]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint32_t gn = bx * BN + wc * 32 * NT + nt * 32 + r;
        if (gn >= N) continue;
        const float bs = Bsq[gn], bi = Binv[gn];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t gm = by * kGemmBM + wr * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (gm >= M) continue;
                out[(uint64_t)gm * ostride + gn] = approx_distance(metric, (acc[mt][nt][e] * Ainv[gm]) * bi, Asq[gm], bs);
            }
    }
}

// The same contraction for a level whose frontiers differ per query: one block per work item of
// pair_scan_kernel = a parent node and up to kGroupPairs (64) of the queries that chose it. A rows are
// gathered by query; B rows are the children's rows of the level table, contiguous because the level
// is laid out breadth-first. d~ goes where the grouped exact kernel puts d:
// fdist[query][position of the parent's first child in the query's frontier + child].
// These are small problems (64 x <= a few hundred x K) and there are thousands of them, so a block's
// time is load latency, not arithmetic: the block keeps a 64-row A chunk and a 64-row B chunk of KC
// values in LDS (small, so that many blocks are in flight per CU), walks the parent's children in
// tiles of 64, and fetches the next chunk while the matrix cores work on the current one. 4 waves as
// 2 x 2, each one 32 x 32 MFMA tile (two accumulators, see split_gemm_tile).
// grid = bound on the items, block 256, LDS route_gemm_grouped_lds_bytes<KC>(). kp a multiple of KC.
template <int KC>
constexpr size_t route_gemm_grouped_lds_bytes() { return (size_t)2 * 64 * (KC / 4 + 1) * 16; }

template <int KC>
__global__ void __launch_bounds__(256, 2)
route_gemm_grouped_kernel(IndexDev ix, const u32x4 *A, const float *Asq, const u32x4 *B, const float *Bsq, uint32_t kp,
                          uint32_t n_parents, const uint32_t *ioff, const uint4 *items, const uint32_t *sorted, float *fdist,
                          uint32_t fstride) {
    constexpr int CHU = KC / 4, RS = CHU + 1, U = CHU / 4; // 16-byte units per chunk row; LDS row stride; units per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *As = reinterpret_cast<u32x4 *>(smem); // [64][RS]
    u32x4 *Bs = As + 64 * RS;                    // [64][RS]
    __shared__ uint32_t pk[64];
    __shared__ float pasq[64]; // squared norms of the pairs' queries
    const uint32_t item = blockIdx.x;
    const uint32_t n_items = ioff[n_parents];
    const uint4 rec = items[min(item, max(n_items, 1u) - 1)];
    if (item >= n_items) return;
    CUDDLE_STAMP(7, 0);
    const uint32_t pb = rec.x, np = rec.y, cpos0 = rec.z, cc = rec.w; // (pair_scan_kernel's contraction form)
    if (cc == 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    const uint32_t lrow = tid >> 2, lu = tid & 3, rowu = kp / 4, nchunks = kp / KC;
    const uint32_t mypair = sorted[pb + min(lrow, np - 1)]; // unused rows repeat the last pair
    if (lu == 0) {
        pk[lrow] = mypair;
        pasq[lrow] = Asq[mypair >> kPairOffBits];
    }
    const u32x4 *ga = A + (uint64_t)(mypair >> kPairOffBits) * rowu + lu;
    const u32x4 *Bp = B + (uint64_t)cpos0 * rowu + lu;
    u32x4 pa[U], pbv[U];
    const uint32_t ntiles = (cc + 63) / 64, steps = ntiles * nchunks;
    auto fetch = [&](uint32_t step) { // step = column tile * nchunks + chunk
        const uint32_t bx = step / nchunks, ch = step - bx * nchunks;
        const u32x4 *gb = Bp + (uint64_t)min(bx * 64 + lrow, cc - 1) * rowu + ch * CHU;
#pragma unroll
        for (int j = 0; j < U; ++j) pbv[j] = gb[4 * j];
        if (nchunks > 1 || step == 0) {
#pragma unroll
            for (int j = 0; j < U; ++j) pa[j] = ga[ch * CHU + 4 * j];
        }
    };
    f32x16 acc, cor;
    fetch(0);
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t bx = step / nchunks, ch = step - bx * nchunks;
        if (ch == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = cor[e] = 0.f;
        }
        __syncthreads(); // the previous step's LDS reads are done (first step: pk written)
#pragma unroll
        for (int j = 0; j < U; ++j) Bs[lrow * RS + lu + 4 * j] = pbv[j];
        if (nchunks > 1 || step == 0) {
#pragma unroll
            for (int j = 0; j < U; ++j) As[lrow * RS + lu + 4 * j] = pa[j];
        }
        __syncthreads();
        if (step == 0) CUDDLE_STAMP(7, 1);
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const uint32_t g = 2 * (2 * ks + h);
            const u32x4 *pa4 = As + (wr * 32 + r) * RS + g, *pb4 = Bs + (wc * 32 + r) * RS + g;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, pa4[0]), al = __builtin_bit_cast(bf16x8, pa4[1]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, pb4[0]), bl = __builtin_bit_cast(bf16x8, pb4[1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, cor, 0, 0, 0);
            cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, cor, 0, 0, 0);
        }
        if (step == 0) CUDDLE_STAMP(7, 2);
        if (ch + 1 == nchunks) { // the tile is complete: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
            const uint32_t gn = bx * 64 + wc * 32 + r;
            if (gn < cc) {
                const float bs = Bsq[cpos0 + gn];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t prow = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (prow >= np) continue;
                    const uint32_t v = pk[prow], q = v >> kPairOffBits, off = v & ((1u << kPairOffBits) - 1);
                    fdist[(uint64_t)q * fstride + off + gn] = approx_distance(ix.metric, acc[e] + cor[e], pasq[prow], bs);
                }
            }
        }
        if (step == 0) CUDDLE_STAMP(7, 3);
    }
    CUDDLE_STAMP(7, 4);
}

// route_gemm_grouped_kernel from the single-pass half-precision operands (center_queries16_kernel /
// level_table16_kernel): one matrix-core product per value pair, half the bytes per row. Same work items, same
// output places.
template <int KC>
constexpr size_t route_gemm16_grouped_lds_bytes() { return (size_t)2 * 64 * (KC / 8 + 1) * 16; }

template <int KC>
__global__ void __launch_bounds__(256, 2)
route_gemm16_grouped_kernel(IndexDev ix, const u32x4 *A, const float *Asq, const float *Ainv, const u32x4 *B, const float *Bsq,
                            const float *Binv, uint32_t kp, uint32_t n_parents, const uint32_t *ioff, const uint4 *items,
                            const uint32_t *sorted, float *fdist, uint32_t fstride) {
    constexpr int CHU = KC / 8, RS = CHU + 1, U = CHU / 4; // 16-byte units per chunk row; LDS row stride; units per thread
    static_assert(CHU % 4 == 0, "four threads share a row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *As = reinterpret_cast<u32x4 *>(smem); // [64][RS]
    u32x4 *Bs = As + 64 * RS;                    // [64][RS]
    __shared__ uint32_t pk[64];
    __shared__ float pasq[64], painv[64]; // squared norms and inverse scales of the pairs' queries
    const uint32_t item = blockIdx.x;
    const uint32_t n_items = ioff[n_parents];
    const uint4 rec = items[min(item, max(n_items, 1u) - 1)];
    if (item >= n_items) return;
    const uint32_t pb = rec.x, np = rec.y, cpos0 = rec.z, cc = rec.w; // (pair_scan_kernel's contraction form)
    if (cc == 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    const uint32_t lrow = tid >> 2, lu = tid & 3, rowu = kp / 8, nchunks = kp / KC;
    const uint32_t mypair = sorted[pb + min(lrow, np - 1)]; // unused rows repeat the last pair
    if (lu == 0) {
        pk[lrow] = mypair;
        pasq[lrow] = Asq[mypair >> kPairOffBits];
        painv[lrow] = Ainv[mypair >> kPairOffBits];
    }
    const u32x4 *ga = A + (uint64_t)(mypair >> kPairOffBits) * rowu + lu;
    const u32x4 *Bp = B + (uint64_t)cpos0 * rowu + lu;
    u32x4 pa[U], pbv[U];
    const uint32_t ntiles = (cc + 63) / 64, steps = ntiles * nchunks;
    auto fetch = [&](uint32_t step) { // step = column tile * nchunks + chunk
        const uint32_t bx = step / nchunks, ch = step - bx * nchunks;
        const u32x4 *gb = Bp + (uint64_t)min(bx * 64 + lrow, cc - 1) * rowu + ch * CHU;
#pragma unroll
        for (int j = 0; j < U; ++j) pbv[j] = gb[4 * j];
        if (nchunks > 1 || step == 0) {
#pragma unroll
            for (int j = 0; j < U; ++j) pa[j] = ga[ch * CHU + 4 * j];
        }
    };
    f32x16 acc;
    fetch(0);
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t bx = step / nchunks, ch = step - bx * nchunks;
        if (ch == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        }
        __syncthreads(); // the previous step's LDS reads are done (first step: pk written)
#pragma unroll
        for (int j = 0; j < U; ++j) Bs[lrow * RS + lu + 4 * j] = pbv[j];
        if (nchunks > 1 || step == 0) {
#pragma unroll
            for (int j = 0; j < U; ++j) As[lrow * RS + lu + 4 * j] = pa[j];
        }
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const uint32_t g = 2 * ks + h;
            const f16x8 a = __builtin_bit_cast(f16x8, As[(wr * 32 + r) * RS + g]);
            const f16x8 b = __builtin_bit_cast(f16x8, Bs[(wc * 32 + r) * RS + g]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (ch + 1 == nchunks) { // the tile is complete: C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
            const uint32_t gn = bx * 64 + wc * 32 + r;
            if (gn < cc) {
                const float bs = Bsq[cpos0 + gn], bi = Binv[cpos0 + gn];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t prow = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (prow >= np) continue;
                    const uint32_t v = pk[prow], q = v >> kPairOffBits, off = v & ((1u << kPairOffBits) - 1);
                    fdist[(uint64_t)q * fstride + off + gn] = approx_distance(ix.metric, (acc[e] * painv[prow]) * bi, pasq[prow], bs);
                }
            }
        }
    }
}

// Bound on |d~ - d| (d = the exact-order f32 distance) for one query against any node of the
// level: any-order f32 dot products of length D are within gamma_D * sum|x_i y_i| of the true
// value, the exact-order L2 within gamma_(D/16+8) of it, centring adds u-relative component
// errors, and the split operands drop at most 3.0001 * 2^-16 sum|x_i y_i|; collected conservatively
// as ((2.5 D + 64) * 2^-24 + 3.25 * 2^-16) times
//   L2: |q'|^2 + max|c'|^2      DOT: |q| max|c|      COSINE: 1.
// (L2: the error of S enters d~ twice, and 2 sum|x_i y_i| <= |q'|^2 + |c'|^2.)
// half != 0: the single-pass half-precision operands, whose per-product term is 1.01 * 2^-10 instead
// (see level_table16_kernel; D <= 32768, so 2^-26 sqrt(D) is below a hundredth of 2^-10).
__device__ __forceinline__ float route_error_bound(uint32_t metric, uint32_t dim, float qsq, float max_csq, uint32_t half = 0) {
    const float cu = (2.5f * (float)dim + 64.0f) * 5.9604645e-8f + (half ? 1.01f * 9.765625e-4f : 3.25f * 1.52587890625e-5f);
    if (metric == kMetricL2) return cu * (qsq + max_csq);
    if (metric == kMetricDot) return cu * sqrtf(qsq) * sqrtf(max_csq);
    return cu;
}

// Diagnostic: how close the approximate distances come to the bound the exactness proof relies on.
// For every (query, node of the level): err = |d~ - d| with d evaluated in f64 from the original
// vectors; dmax[0] = max err / route_error_bound(query), dmax[1] = max err (bits of non-negative
// floats, so the unsigned maximum orders them). grid = (ceil(n/256), nq), block 256.
__global__ void __launch_bounds__(256)
route_filter_error_kernel(IndexDev ix, const float *queries, const uint32_t *level_nodes, uint32_t n, const float *fdist,
                          uint32_t fstride, const float *qsq, float max_csq, uint32_t *dmax, uint32_t half) {
    const uint32_t q = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, D = ix.dim;
    if (j >= n) return;
    const float *a = queries + (uint64_t)q * D, *b = ix.centroids + (uint64_t)level_nodes[j] * D;
    double dot = 0.0, aa = 0.0, bb = 0.0, l2 = 0.0;
    for (uint32_t i = 0; i < D; ++i) {
        const double x = a[i], y = b[i];
        dot += x * y;
        aa += x * x;
        bb += y * y;
        l2 += (x - y) * (x - y);
    }
    double d;
    if (ix.metric == kMetricL2) d = l2;
    else if (ix.metric == kMetricDot) d = -dot;
    else d = (aa == 0.0 || bb == 0.0) ? 1.0 : 1.0 - dot / (sqrt(aa) * sqrt(bb));
    const float err = (float)fabs((double)fdist[(uint64_t)q * fstride + j] - d);
    const float bound = route_error_bound(ix.metric, D, qsq[q], max_csq, half);
    atomicMax(&dmax[0], __float_as_uint(bound > 0.f ? err / bound : (err > 0.f ? 3.4e38f : 0.f)));
    atomicMax(&dmax[1], __float_as_uint(err));
}

// Per query: the kc smallest approximate distances of the level (positions in level order) and
// the kc-th smallest value itself. One block per query.
__global__ void __launch_bounds__(256)
route_prefilter_kernel(const float *fdist, uint32_t fstride, uint32_t n, uint32_t kc, uint32_t *cand_pos,
                       uint32_t cstride, float *cand_max) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t misc[2];
    __shared__ uint32_t wtmp[32];
    const uint32_t q = blockIdx.x;
    const float *fd = fdist + (uint64_t)q * fstride;
    uint32_t budget;
    const uint32_t T = block_radix_select(n, kc, [&](uint32_t i) { return f32_key(fd[i]); }, hist, misc, budget);
    uint32_t *cp = cand_pos + (uint64_t)q * cstride;
    ordered_threshold_compact(
        n, T, budget, [&](uint32_t i) { return f32_key(fd[i]); },
        [&](uint32_t i, uint32_t, uint32_t rank) { if (rank < cstride) cp[rank] = i; }, wtmp);
    if (threadIdx.x == 0) cand_max[q] = f32_from_key(T);
}

// Same result as route_prefilter_kernel for levels of at most KPT * THREADS nodes, with
// the keys read once and kept in registers: block min/max first (radix passes whose digit is the
// same for every key are skipped: no counting, no same-address atomics), then the remaining
// passes from registers; the winners are written in any order (which ties at the threshold are
// kept is free: every unselected node still has d~ >= cand_max, which is all the proof needs).
// ncount != null: per-query frontiers of ncount[q] nodes (at most n); queries with skip[q] != 0 are
// left alone; cand_cnt[q] = the query's number of candidates, min(kc, ncount[q]).
template <int THREADS, int KPT>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(KPT <= 20 ? 8 : 4)))
route_prefilter_reg_kernel(const float *fdist, uint32_t fstride, uint32_t n, uint32_t kc, uint32_t *cand_pos,
                           uint32_t cstride, float *cand_max, const uint32_t *ncount = nullptr,
                           const uint32_t *skip = nullptr, uint32_t *cand_cnt = nullptr) {
    __shared__ uint32_t hist[4 * 256];
    __shared__ uint32_t misc[4];
    __shared__ uint32_t wred[2 * (THREADS / 64)];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    CUDDLE_STAMP(8, 6);
    if (skip && skip[q]) return;
    if (ncount) { // per-query frontier: its own width, and no more candidates than it has nodes
        n = min(n, ncount[q]);
        kc = min(kc, n);
        if (tid == 0) cand_cnt[q] = kc;
        if (n == 0) return;
    }
    const float *fd = fdist + (uint64_t)q * fstride;
    uint32_t keys[KPT];
    // every load is issued before the first value is used (a load behind a condition is waited for on its own: twenty
    // trips to the L2, one after the other); positions past the frontier re-read its last key and are masked afterwards
    float kv[KPT];
#pragma unroll
    for (int t = 0; t < KPT; ++t) kv[t] = fd[min((uint32_t)t * THREADS + tid, n - 1)];
#pragma unroll
    for (int t = 0; t < KPT; ++t) keys[t] = (uint32_t)t * THREADS + tid < n ? f32_key(kv[t]) : 0xFFFFFFFFu;
    if (tid < 3) misc[1 + tid] = 0; // shortlist and winner counters (visible after the barriers inside the select)
    uint32_t *cp = cand_pos + (uint64_t)q * cstride;
    CUDDLE_STAMP(8, 0);
    // Few candidates out of many keys: the kc-th smallest of the threads' own minima is an upper bound
    // T0 of the kc-th smallest key (kc minima, hence kc keys, are <= T0), and only a few more than kc
    // keys pass it. Selecting among the THREADS minima and then among that shortlist costs two cheap
    // selections instead of counting sweeps over all the keys.
    constexpr uint32_t SHORT = 2 * THREADS;
    __shared__ uint32_t skey[KPT >= 8 ? SHORT : 1];
    uint32_t T = 0, budget = 0;
    bool have = false;
    if (KPT >= 8 && kc * 2 <= min(n, (uint32_t)THREADS)) { // (then about 1.4 kc keys pass T0)
        uint32_t lmin[1] = {keys[0]};
#pragma unroll
        for (int t = 1; t < KPT; ++t) lmin[0] = min(lmin[0], keys[t]);
        uint32_t b0;
        CUDDLE_STAMP(8, 1);
        // (a cut, not the exact kc-th minimum: the selection stops at the digit that leaves at most four minima in its
        // bucket and takes the bucket's upper edge)
        const uint32_t T0 = reg_radix_select<THREADS, 1, true>(lmin, min(n, (uint32_t)THREADS), kc, hist, wred, b0);
        CUDDLE_STAMP(8, 2);
        if (lmin[0] <= T0) {
#pragma unroll
            for (int t = 0; t < KPT; ++t)
                if (keys[t] <= T0 && (uint32_t)t * THREADS + tid < n) {
                    const uint32_t slot = atomicAdd(&misc[1], 1u);
                    if (slot < SHORT) skey[slot] = keys[t];
                }
        }
        __syncthreads(); // also: every wave is done with the first selection's counters
        const uint32_t m = misc[1];
        CUDDLE_STAMP(8, 3);
        if (m <= SHORT) { // (more: many equal keys; the plain selection below handles any input)
            uint32_t k2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t j = t * THREADS + tid;
                k2[t] = j < m ? skey[j] : 0xFFFFFFFFu;
            }
            T = reg_radix_select<THREADS, 2>(k2, m, kc, hist, wred, budget);
            have = true;
        }
        __syncthreads(); // the selection's counters may be reused below
    }
    if (!have) T = reg_radix_select<THREADS, KPT>(keys, n, kc, hist, wred, budget);
    CUDDLE_STAMP(8, 4);
    // winners from the registers, i.e. roughly in level order: neighbouring candidates are neighbouring
    // rows of the centroid array, which the exact refinement reads next
    const uint32_t n_lt = kc - budget;
#pragma unroll
    for (int t = 0; t < KPT; ++t) {
        const uint32_t i = t * THREADS + tid;
        if (i < n) {
            if (keys[t] < T) {
                const uint32_t slot = atomicAdd(&misc[2], 1u);
                if (slot < cstride) cp[slot] = i;
            } else if (keys[t] == T) {
                const uint32_t ts = atomicAdd(&misc[3], 1u);
                if (ts < budget && n_lt + ts < cstride) cp[n_lt + ts] = i;
            }
        }
    }
    if (tid == 0) cand_max[q] = f32_from_key(T);
    CUDDLE_STAMP(8, 5);
}

// Exact distance(query, centroid, metric) of the prefiltered nodes (levels on this path hold more
// than 100 nodes, so search() uses distance() for them, hierarchical.rs:355-365).
// cand_cnt != null: per-query frontiers (level_nodes = the frontier array, fstride per query) with
// cand_cnt[q] candidates; queries with skip[q] != 0 are left alone.
// grid = (ceil(kc/64), nq), block 256: one 4-lane group per (query, candidate).
__global__ void __launch_bounds__(256)
route_refine_kernel(IndexDev ix, const float *queries, const uint32_t *level_nodes, const uint32_t *cand_pos,
                    uint32_t cstride, uint32_t kc, float *cand_dist, const float *qnorm, uint32_t fstride = 0,
                    const uint32_t *skip = nullptr, const uint32_t *cand_cnt = nullptr, const uint32_t *seg = nullptr,
                    uint32_t pstride = 0, uint32_t *zero = nullptr, uint32_t n_zero = 0, uint32_t bucket_div = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);
    uint32_t *st = reinterpret_cast<uint32_t *>(qs + ((ix.dim + 3) & ~3u)); // the query's segment table (seg != null), or the block's candidates (bucket_div != 0)
    const uint32_t q = blockIdx.y, D = ix.dim;
    if (zero) { // the pair counters the level's finish kernel fills next (instead of a memset launch of their own)
        const uint32_t nt = gridDim.x * gridDim.y * blockDim.x; // (the blocks of z == 0 are enough)
        if (blockIdx.z == 0)
            for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n_zero; i += nt) zero[i] = 0;
    }
    if (skip && skip[q]) return;
    if (cand_cnt) { // per-query frontiers: level_nodes is the array of frontiers, fstride nodes per query
        kc = cand_cnt[q];
        level_nodes += (uint64_t)q * fstride;
        if (blockIdx.x * 64 >= kc) return;
    }
    const float *qg = queries + (uint64_t)q * D;
    for (uint32_t i = threadIdx.x; i < D; i += blockDim.x) qs[i] = qg[i];
    if (seg)
        for (uint32_t i = threadIdx.x; i < seg_stride(pstride); i += blockDim.x) st[i] = seg[(uint64_t)q * seg_stride(pstride) + i];
    if (bucket_div) {
        // Shared level, grid (8, nq, Z): block (x, q, z) takes the query's candidates in the z-th part of the x-th eighth
        // of the level (bucket_div nodes per part). Workgroups go to the 8 XCDs in turn by linear number, so the blocks
        // x run on XCD x, whose L2 then serves one eighth of the level's centroid rows to every query of the batch
        // instead of a share of all of them - and, z being the slowest grid dimension, one part of the eighth at a time
        // (Z is chosen so that a part fits the L2). Each block looks through the query's candidate list for its own.
        __shared__ uint32_t n_mine;
        if (threadIdx.x == 0) n_mine = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < kc; j += blockDim.x)
            if (cand_pos[(uint64_t)q * cstride + j] / bucket_div == blockIdx.x * gridDim.z + blockIdx.z) st[atomicAdd(&n_mine, 1u)] = j;
        __syncthreads();
        const uint32_t cnt = n_mine, sub = threadIdx.x & 3;
        for (uint32_t e = threadIdx.x >> 2; e < cnt; e += blockDim.x >> 2) { // (whole 4-lane groups leave: see rerank_kernel)
            const uint32_t j = st[e], node = level_nodes[cand_pos[(uint64_t)q * cstride + j]];
            const float d = distance_group4_qn(qs, ix.centroids + (uint64_t)node * D, D, ix.metric, sub, (D & 3) == 0,
                                               ix.metric == kMetricCosine ? qnorm[q] : 0.f);
            if (sub == 0) cand_dist[(uint64_t)q * cstride + j] = d;
        }
        return;
    }
    __syncthreads();
    const uint32_t j = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    if (j >= kc) return; // (whole 4-lane groups leave: see rerank_kernel)
    const uint32_t cpos = cand_pos[(uint64_t)q * cstride + j];
    const uint32_t node = seg ? frontier_node(ix, st, pstride, cpos) : level_nodes[cpos];
    float d = distance_group4_qn(qs, ix.centroids + (uint64_t)node * D, D, ix.metric, sub, (D & 3) == 0,
                                 ix.metric == kMetricCosine ? qnorm[q] : 0.f);
    if (sub == 0) cand_dist[(uint64_t)q * cstride + j] = d;
}

inline size_t route_finish_lds_bytes(uint32_t kpad, uint32_t probes) { // keys, expansion scratch, segment table
    return (size_t)kpad * 8 + 4 * ((size_t)probes + 1) * 4 + (size_t)seg_stride(probes) * 4 + 16;
}

// Per query: sort the refined candidates by (exact distance, level position); the first `probes`
// are the exact selection provided no unselected node can beat the probes-th one:
//     cand_max - error_bound > exact distance of the probes-th      (or every node was a candidate).
// Then expand as route_select_kernel does. Otherwise raise flag[q] for the exact path.
// cand_cnt != null: per-query frontiers (level_nodes = the frontier array, fstride per query) of
// ncount[q] nodes with cand_cnt[q] >= probes candidates; kc = the largest candidate count (LDS size).
// Queries with skip[q] != 0 are flagged for the exact path without being counted as fallbacks.
// One block per query; dynamic LDS = route_finish_lds_bytes(next_pow2(kc), probes).
__global__ void __launch_bounds__(256)
route_finish_kernel(IndexDev ix, const uint32_t *level_nodes, uint32_t n, const uint32_t *cand_pos,
                    const float *cand_dist, uint32_t cstride, uint32_t kc, const float *cand_max, const float *qsq,
                    float max_csq, uint32_t probes, uint32_t *next_frontier, uint32_t fstride, uint32_t *next_fcount,
                    uint32_t *leaves, uint32_t lstride, uint32_t *lcount, uint32_t *flag, uint32_t *fallbacks, PairOut po,
                    const uint32_t *ncount = nullptr, const uint32_t *skip = nullptr, const uint32_t *cand_cnt = nullptr,
                    const uint32_t *seg = nullptr, uint32_t half_filter = 0, uint32_t *half_fallbacks = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t kpad = next_pow2(kc);
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);
    uint32_t *xs = reinterpret_cast<uint32_t *>(skeys + kpad);
    uint32_t *st = xs + 4 * (probes + 1); // this level's segment table of the query (seg != null)
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (skip && skip[q]) { // not on this path (small frontier): the exact kernels take the query
        if (tid == 0) flag[q] = 1;
        return;
    }
    if (cand_cnt) { // per-query frontiers: level_nodes is the array of frontiers, fstride nodes per query
        n = ncount[q];
        kc = cand_cnt[q];
        level_nodes += (uint64_t)q * fstride;
    }
    CUDDLE_STAMP(9, 0);
    if (seg)
        for (uint32_t i = tid; i < seg_stride(probes); i += blockDim.x) st[i] = seg[(uint64_t)q * seg_stride(probes) + i];
    if (kpad <= 256) {
        // few keys, all distinct (the position is part of the key): every thread finds its key's rank by
        // counting the smaller ones, which takes one pass over LDS and two barriers
        const uint64_t mine = tid < kc ? (((uint64_t)f32_key(cand_dist[(uint64_t)q * cstride + tid]) << 32) | cand_pos[(uint64_t)q * cstride + tid])
                                       : ~0ull;
        if (tid < kpad) skeys[tid] = mine;
        __syncthreads();
        CUDDLE_STAMP(9, 1);
        uint32_t rank = 0;
        if (tid < kc)
            for (uint32_t j = 0; j < kc; ++j) rank += skeys[j] < mine ? 1u : 0u;
        __syncthreads();
        if (tid < kc) skeys[rank] = mine;
        __syncthreads();
    } else {
        for (uint32_t i = tid; i < kpad; i += blockDim.x)
            skeys[i] = i < kc ? (((uint64_t)f32_key(cand_dist[(uint64_t)q * cstride + i]) << 32) | cand_pos[(uint64_t)q * cstride + i])
                              : ~0ull;
        __syncthreads();
        CUDDLE_STAMP(9, 1);
        block_sort_u64(skeys, kpad);
    }
    CUDDLE_STAMP(9, 2);
    const float tau = f32_from_key((uint32_t)(skeys[probes - 1] >> 32));
    const float e = route_error_bound(ix.metric, ix.dim, qsq[q], max_csq, half_filter);
    const bool ok = kc >= n || (cand_max[q] - e > tau);
    if (!ok) {
        if (tid == 0) {
            flag[q] = 1;
            atomicAdd(fallbacks, 1u);
            if (half_fallbacks) atomicAdd(half_fallbacks, 1u); // this level's count, never reset: the host judges the filter by it
        }
        return;
    }
    if (tid == 0) flag[q] = 0;
    if (seg)
        route_expand_fn(ix, [&](uint32_t pos) { return frontier_node(ix, st, probes, pos); }, skeys, probes, xs, probes + 1,
                        next_frontier + (uint64_t)q * fstride, fstride, next_fcount + q, leaves + (uint64_t)q * lstride, lstride,
                        lcount + q, po, q);
    else
        route_expand(ix, level_nodes, skeys, probes, xs, probes + 1, next_frontier + (uint64_t)q * fstride, fstride,
                     next_fcount + q, leaves + (uint64_t)q * lstride, lstride, lcount + q, po, q);
    CUDDLE_STAMP(9, 3);
}

} // namespace cuddle
