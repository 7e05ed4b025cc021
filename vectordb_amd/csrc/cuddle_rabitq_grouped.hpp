// cuddle_rabitq_grouped.hpp — the bound half of RaBitQ phase 1 on the matrix cores, leaf by leaf.
//
// The bound of cuddle_rabitq.hpp needs, for every (query, candidate) pair, the integer
//     S = sum_i (bit_i ? +1 : -1) * sgn(rq_i) * v_i           (v_i = the query's magnitudes quantised to kRqPlanes bits)
// and a leaf's code bits are the same for every query that reaches the leaf. So a batch is turned round: the
// (query, leaf) pairs of all leaf lists are sorted by leaf, and one block takes a leaf with up to kRqItemPairs of
// its queries. The leaf's code bits are expanded to bytes (0 / 1) once, the queries' values travel as two signed
// 5-bit pieces (v = 32 hi + lo, both with the sign of rq_i), and
//     P = sum_i bit_i * r_i = 32 * (bits x hi) + (bits x lo)  on v_mfma_i32_16x16x64_i8,    S = 2 P - sum_i r_i,
// in exact integer arithmetic: the same S as the popcount form, hence the same bounds, the same threshold, the
// same candidates evaluated exactly afterwards (reference: estimate_distance_fast, src/quantization/rabitq.rs:335-363,
// evaluated for every candidate of collect_leaf_candidates, src/index/hierarchical_with_rabitq.rs:605-619).
// The per-query kernel (scan_select_rabitq_kernel with pre_bounds) then starts from the bounds this leaves in the
// query's row of the spill buffer.
#pragma once

#include "cuddle_rabitq.hpp"

namespace cuddle {

constexpr uint32_t kRqItemPairs = 64; // (query, leaf) pairs per work item = kGroupPairs (the pair kernels cut items of that size)
static_assert(kRqItemPairs == (uint32_t)kGroupPairs, "work items are cut by pair_scan_kernel");

// Bytes per query of the signed 5-bit pieces: [2][dim rounded up to 64] i8, piece 0 = low, piece 1 = high.
__host__ __device__ inline uint32_t rq_i8_stride(uint32_t dim) { return 2 * ((dim + 63) & ~63u); }

// One pair per non-empty reached leaf of every query: (query, scan position of the leaf's first vector in the query's
// candidate list), its leaf (node id) and its place among the leaf's pairs. tables: the blocks' leaf tables of
// scan_tables_kernel (row 0 ends with the query number and its leaf count). grid = (ceil(lstride / 256), nq).
__global__ void rq_pairs_kernel(const uint32_t *tables, uint32_t tstride, const uint32_t *leaves, uint32_t lstride, uint32_t *cnt,
                                uint2 *info, uint32_t *node_of, uint32_t *slot_of) {
    const uint32_t b = blockIdx.y, sl = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *tb = tables + (uint64_t)b * 4 * tstride;
    const uint32_t q = tb[tstride - 2], nl = min(tb[tstride - 1], lstride);
    if (sl >= lstride) return;
    const uint64_t at = (uint64_t)b * lstride + sl;
    uint2 v = make_uint2(0xFFFFFFFFu, 0u);
    uint32_t node = 0, place = 0;
    if (sl < nl) {
        const uint32_t base = tb[tstride + sl];
        if (tb[tstride + sl + 1] > base) {
            node = leaves[(uint64_t)q * lstride + sl];
            place = atomicAdd(&cnt[node], 1u);
            v = make_uint2(q, base);
        }
    }
    info[at] = v;
    node_of[at] = node;
    slot_of[at] = place;
}

// The pairs in leaf order (order inside a leaf is immaterial).
__global__ void rq_pair_fill_kernel(const uint2 *info, const uint32_t *node_of, const uint32_t *slot_of, uint32_t total,
                                    const uint32_t *poff, uint2 *sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint2 v = info[i];
    if (v.x == 0xFFFFFFFFu) return;
    sorted[poff[node_of[i]] + slot_of[i]] = v;
}

// One WAVE per work item (leaf, up to kRqItemPairs of the queries that reach it); four items per block, waves past
// the last item exit. The item's queries go by in groups of 16 (the columns of the 16x16x64 product); for a group
// the wave walks the K steps of 64 dimensions once, keeping the integer accumulators of up to MT 16-vector tiles of
// the leaf in registers: per step the queries' two pieces are loaded once (16 bytes per lane and piece) and every
// tile's 16 code bits per lane are expanded to bytes and multiplied twice. The next step's operands are requested
// before the current step is computed.
//   items: (leaf node, first pair, pairs, -) from pair_scan_kernel; sorted: (query, scan base) of the pairs in leaf order
//   qi8: [nq][rq_i8_stride] signed pieces; qrows: the queries' rows of rabitq_query_planes_kernel (scalars)
//   gest: [nq][2 * gest_stride] (upper bounds | lower bounds)
template <int MT>
__global__ void __launch_bounds__(256)
rq_bounds_grouped_kernel(IndexDev ix, const uint32_t *n_items, const uint4 *items, const uint2 *sorted, const int8_t *qi8,
                         const uint32_t *qrows, const float *qnorm_all, float *gest, uint32_t gest_stride) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= *n_items) return;
    const uint4 it = items[item];
    const uint32_t leaf = it.x, first = it.y, npairs = it.z;
    const uint32_t W32 = ix.words32, D = ix.dim, kpad = (D + 63) & ~63u, nk = kpad >> 6;
    const uint32_t beg = ix.vec_off[leaf], n = ix.vec_off[leaf + 1] - beg;
    const uint32_t *codes = ix.rq_codes + (uint64_t)beg * W32;
    const uint32_t r = lane & 15, kq = lane >> 4;
    struct Step { v4i blo, bhi; uint32_t cw[MT]; };
    for (uint32_t g0 = 0; g0 < npairs; g0 += 16) {
        // column r of this group: its query and where the leaf's candidates start in that query's candidate list
        const bool have = g0 + r < npairs;
        const uint2 pr = sorted[first + (have ? g0 + r : 0)];
        const uint32_t q = pr.x, base = pr.y;
        const uint32_t *row = qrows + (uint64_t)q * rq_plane_stride(W32);
        const float dl = __uint_as_float(row[0]), aq = __uint_as_float(row[1]), bq = __uint_as_float(row[2]), qn = qnorm_all[q];
        const int vs = (int)row[4];
        const int8_t *bp = qi8 + (uint64_t)q * 2 * kpad + 16 * kq;
        float *ge = gest + (uint64_t)q * 2 * gest_stride + base, *glb = ge + gest_stride;
        for (uint32_t m0 = 0; m0 < n; m0 += 16 * MT) {
            auto load = [&](uint32_t s, Step &st) {
                // (columns without a pair read column 0's pieces: their results are never stored)
                st.blo = *reinterpret_cast<const v4i *>(bp + 64 * s);
                st.bhi = *reinterpret_cast<const v4i *>(bp + kpad + 64 * s);
                const uint32_t wd = min(2 * s + (kq >> 1), W32 - 1);
#pragma unroll
                for (int t = 0; t < MT; ++t) st.cw[t] = codes[(uint64_t)wd * n + min(m0 + 16 * t + r, n - 1)];
            };
            v4i acc_lo[MT], acc_hi[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) { acc_lo[t] = v4i{0, 0, 0, 0}; acc_hi[t] = v4i{0, 0, 0, 0}; }
            Step cur, nxt;
            load(0, cur);
            for (uint32_t s = 0; s < nk; ++s) {
                load(min(s + 1, nk - 1), nxt);
                const bool word_in = 2 * s + (kq >> 1) < W32; // (a last step of 32 dimensions has no second word)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const uint32_t h = word_in ? (cur.cw[t] >> (16 * (kq & 1))) & 0xFFFFu : 0u;
                    v4i a; // bit j of h -> byte j (0 / 1): (4 bits x 0x204081) & 0x01010101 puts bit b at bit 8 b
                    a.x = (int)(__umul24(h & 15u, 0x204081u) & 0x01010101u);
                    a.y = (int)(__umul24((h >> 4) & 15u, 0x204081u) & 0x01010101u);
                    a.z = (int)(__umul24((h >> 8) & 15u, 0x204081u) & 0x01010101u);
                    a.w = (int)(__umul24(h >> 12, 0x204081u) & 0x01010101u);
                    acc_lo[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cur.blo, acc_lo[t], 0, 0, 0);
                    acc_hi[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cur.bhi, acc_hi[t], 0, 0, 0);
                }
                cur = nxt;
            }
            // results: lane (column r = query, rows 4 kq .. 4 kq + 3 of tile t = vectors m0 + 16 t + 4 kq + j)
            if (have) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t i = m0 + 16 * t + 4 * kq + j;
                        if (i < n) {
                            const uint32_t pos = beg + i;
                            float lb;
                            const float ub = rabitq_bounds_from_sum(2 * (32 * acc_hi[t][j] + acc_lo[t][j]) - vs, dl, aq, bq, qn,
                                                                    ix.rq_norm[pos], ix.rq_ipoo[pos], lb);
                            ge[i] = ub;
                            glb[i] = lb;
                        }
                    }
                }
            }
        }
    }
}

} // namespace cuddle
