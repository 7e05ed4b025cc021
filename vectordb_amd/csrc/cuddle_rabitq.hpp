// cuddle_rabitq.hpp — RaBitQ phase 1 (reference: src/quantization/rabitq.rs:191-201, 298-363,
// 398-436; driver src/index/hierarchical_with_rabitq.rs:575-633).
//
// All sums run in the reference's order (strictly sequential f32 adds of separately rounded
// products), so estimates are bit-identical to the reference arithmetic.
#pragma once

#include "cuddle_route.hpp"

namespace cuddle {

// rot[j][i] = rotation_matrix_t[i][j]: the layout in which consecutive lanes (consecutive output
// rows i) read consecutive addresses while every lane walks j in order.
__global__ void transpose_square_kernel(const float *src, float *dst, uint32_t d) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < d && j < d) dst[(uint64_t)j * d + i] = src[(uint64_t)i * d + j];
}

constexpr int kPrepQ = 4; // queries per block of rabitq_query_prep_kernel

// Query half of estimate_distances_batch_fast (rabitq.rs:398-420): qnorm = sqrt(sum q_j^2)
// (sequential), qn = q / qnorm, rq[i] = sum_j rot_t[i][j] * qn[j] (rotate_query, :191-201,
// sequential over j). One block handles kPrepQ queries so each rotation element is loaded once
// for several queries; thread i owns output row i. rq is left untouched for qnorm < 1e-10.
// Dynamic LDS: kPrepQ * dim floats.
__global__ void __launch_bounds__(256)
rabitq_query_prep_kernel(const float *queries, const float *rot, uint32_t nq, uint32_t dim, float *rq, float *qnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qn = reinterpret_cast<float *>(smem); // [kPrepQ][dim]
    __shared__ float s_norm[kPrepQ];
    const uint32_t q0 = blockIdx.x * kPrepQ;
    if (threadIdx.x < kPrepQ) {
        const uint32_t q = q0 + threadIdx.x;
        float s = 0.f;
        if (q < nq)
            for (uint32_t j = 0; j < dim; ++j) {
                float x = queries[(uint64_t)q * dim + j];
                s += x * x;
            }
        float nrm = sqrtf(s);
        s_norm[threadIdx.x] = nrm;
        if (q < nq) qnorm[q] = nrm;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kPrepQ * dim; t += blockDim.x) {
        const uint32_t qq = t / dim, j = t % dim, q = q0 + qq;
        qn[t] = (q < nq && s_norm[qq] >= 1e-10f) ? queries[(uint64_t)q * dim + j] / s_norm[qq] : 0.f;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) {
        float acc[kPrepQ];
#pragma unroll
        for (int qq = 0; qq < kPrepQ; ++qq) acc[qq] = 0.f;
        for (uint32_t j = 0; j < dim; ++j) {
            const float r = rot[(uint64_t)j * dim + i];
#pragma unroll
            for (int qq = 0; qq < kPrepQ; ++qq) acc[qq] += r * qn[qq * dim + j];
        }
#pragma unroll
        for (int qq = 0; qq < kPrepQ; ++qq)
            if (q0 + qq < nq && s_norm[qq] >= 1e-10f) rq[(uint64_t)(q0 + qq) * dim + i] = acc[qq];
    }
}

// estimate_distance_fast (rabitq.rs:335-363) for vector i of a leaf of n vectors whose codes are
// stored word-major ([w][n] u32): ip = sum_i (bit_i ? +rq_i : -rq_i) sequentially
// (compute_inner_product_fast, :298-321), est = (2 - 2 ip / ip_oo) * norm * qnorm.
// The query side arrives split into magnitudes arq[i] = |rq_i| and sign words qsign (bit i of word
// i / 32 = sign bit of rq_i): the sign of term i is sign(rq_i) ^ ~bit_i, i.e. bit i of
// qsign ^ ~code, OR-ed into the magnitude (one and-or instruction) — the same bit pattern as
// flipping rq_i's sign where the code bit is 0, for one operation less per element.
__device__ __forceinline__ float rabitq_estimate(const uint32_t *codes, uint32_t n, uint32_t i, uint32_t dim,
                                                 uint32_t words32, const float *arq, const uint32_t *qsign, float qnorm,
                                                 float norm, float ipoo) {
    if (qnorm < 1e-10f) return norm * norm; // rabitq.rs:408-413 (whole batch) and :342-349
    if (norm < 1e-10f) return qnorm * qnorm;
    // the query side lives in LDS: say so, or the reads are issued as (slower) flat loads
    const LdsF32 *la = (const LdsF32 *)arq;
    const LdsU32 *ls = (const LdsU32 *)qsign;
    float sum = 0.f;
    for (uint32_t w0 = 0; w0 < words32; w0 += 8) { // eight code words in flight per lane
        uint32_t cw[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) cw[u] = w0 + u < words32 ? codes[(uint64_t)(w0 + u) * n + i] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t w = w0 + u;
            if (w >= words32) break;
            const uint32_t sg = ls[w] ^ ~cw[u];
            const uint32_t lim = min(32u, dim - 32u * w);
            if (lim == 32) {
#pragma unroll
                for (uint32_t b = 0; b < 32; ++b)
                    sum += __uint_as_float(((sg << (31 - b)) & 0x80000000u) | __float_as_uint(la[32 * w + b])); // v_and_or_b32
            } else {
                for (uint32_t b = 0; b < lim; ++b)
                    sum += __uint_as_float(((sg << (31 - b)) & 0x80000000u) | __float_as_uint(la[32 * w + b]));
            }
        }
    }
    const float est_ip = sum / ipoo;
    const float dn = 2.0f - 2.0f * est_ip;
    return dn * norm * qnorm;
}

// Stages one query's rotated vector in LDS as magnitudes (arq[dim]) and sign words (qsign); all
// zeros when the query norm is below the reference's cut-off (the estimates then ignore it).
// All threads of the block call; the caller synchronises before reading.
__device__ __forceinline__ void rabitq_stage_query(const float *rq, uint32_t dim, float qnorm, float *arq, uint32_t *qsign) {
    const uint32_t lane = threadIdx.x & 63, dpad = (dim + 63) & ~63u;
    for (uint32_t i = threadIdx.x; i < dpad; i += blockDim.x) { // whole waves: the ballot covers 64 consecutive elements
        const uint32_t bits = (i < dim && qnorm >= 1e-10f) ? __float_as_uint(rq[i]) : 0u;
        if (i < dim) arq[i] = __uint_as_float(bits & 0x7FFFFFFFu);
        const uint64_t sb = __ballot((bits >> 31) != 0);
        if (lane == 0) {
            qsign[i >> 5] = (uint32_t)sb;
            qsign[(i >> 5) + 1] = (uint32_t)(sb >> 32);
        }
    }
}

inline size_t rabitq_scan_lds_bytes(uint32_t dim, uint32_t lcap, uint32_t tcap, uint32_t dist_cap) {
    return scan_lds_bytes(0, 0, lcap, tcap, 0) + 256 * 4 + (size_t)((dim + 3) & ~3u) * 4 + (size_t)((dim + 63) / 64) * 8 +
           (size_t)dist_cap * 4 + 64;
}

struct RabitqLds {
    ScanLds s;
    uint32_t *rhist; // [256] radix-select histogram
    float *rq;       // [dim]  |rotated query|
    uint32_t *qsign; // [2 * ceil(dim / 64)] sign bits of the rotated query
    float *est;      // [dist_cap]
};

__device__ __forceinline__ RabitqLds rabitq_lds_carve(unsigned char *smem, uint32_t dim, uint32_t lcap, uint32_t tcap) {
    RabitqLds r;
    r.s = scan_lds_carve(smem, 0, 0, lcap, tcap);
    uint32_t *p = reinterpret_cast<uint32_t *>(r.s.dist); // end of the common carve (16-byte aligned)
    p = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
    r.rhist = p; p += 256;
    r.rq = reinterpret_cast<float *>(p); p += (dim + 3) & ~3u;
    r.qsign = p; p += 2 * ((dim + 63) / 64);
    p = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
    r.est = reinterpret_cast<float *>(p);
    return r;
}

// One block per query: estimated distance of every vector of every reached leaf
// (compute_rabitq_distances / collect_leaf_candidates, hierarchical_with_rabitq.rs:581-619), then
// the rerank_k smallest by (estimate, scan position) — the first rerank_k of the reference's
// stable sort (select_top_candidates, :623-633) — emitted in scan order with the estimate's bit
// pattern as key. Estimates live in LDS (or the global spill buffer when a query has more than
// dist_cap candidates).
__global__ void __launch_bounds__(512)
scan_select_rabitq_kernel(IndexDev ix, const float *rq_all, const float *qnorm_all, const uint32_t *tables,
                          uint32_t tstride, uint32_t lcap, uint32_t tcap, uint32_t rerank_k,
                          uint32_t dist_cap, float *gest, uint32_t gest_stride, uint32_t *sel_pos, uint32_t *sel_key,
                          uint32_t sel_stride, uint32_t *sel_count, uint32_t *cand_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t D = ix.dim, W32 = ix.words32;
    RabitqLds r = rabitq_lds_carve(smem, D, lcap, tcap);
    const ScanLds &s = r.s;
    CUDDLE_STAMP(6, 0);
    uint32_t ntiles = 0, q, nl;
    scan_copy_tables(s, tables + (uint64_t)blockIdx.x * 4 * tstride, tstride, lcap, &q, &nl);
    const float qnorm = qnorm_all[q];
    rabitq_stage_query(rq_all + (uint64_t)q * D, D, qnorm, r.rq, r.qsign);
    const uint32_t C = scan_build_tiles(s, nl, tcap, &ntiles);
    if (tid == 0) cand_count[q] = C;
    if (C == 0) {
        if (tid == 0) sel_count[q] = 0;
        return;
    }
    const bool use_lds = C <= dist_cap;
    float *ge = gest + (uint64_t)q * gest_stride;
    CUDDLE_STAMP(6, 1);
    for (uint32_t t = w; t < ntiles; t += nw) {
        const uint32_t tile = s.tiles[t], sl = tile >> 12, i = ((tile & 4095u) << 6) + lane;
        const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
        if (i < n) {
            const uint32_t pos = s.lbeg[sl] + i;
            float e = rabitq_estimate(ix.rq_codes + (uint64_t)s.lbeg[sl] * W32, n, i, D, W32, r.rq, r.qsign, qnorm,
                                      ix.rq_norm[pos], ix.rq_ipoo[pos]);
            if (use_lds) r.est[base + i] = e; else ge[base + i] = e;
        }
    }
    __syncthreads();
    CUDDLE_STAMP(6, 2);
    auto key = [&](uint32_t i) -> uint32_t { return f32_key(use_lds ? r.est[i] : ge[i]); };
    auto key2 = [&](uint32_t i, uint32_t &) -> uint32_t { return key(i); };
    uint32_t T = 0xFFFFFFFFu, budget = 0;
    bool all = C <= rerank_k;
    if (!all) T = block_radix_select(C, rerank_k, key, r.rhist, s.misc, budget);
    CUDDLE_STAMP(6, 3);
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    auto emit = [&](uint32_t i, uint32_t, uint32_t, uint32_t rank) {
        if (rank < sel_stride) {
            uint32_t sl = scan_slot_of(s.lscan, nl, i);
            sp[rank] = s.lbeg[sl] + (i - s.lscan[sl]);
            sk[rank] = __float_as_uint(use_lds ? r.est[i] : ge[i]);
        }
    };
    if (all) {
        // every candidate is selected: keys compare below the sentinel except NaN estimates, which
        // the sentinel-equal budget admits as well
        ordered_threshold_compact_aux(C, 0xFFFFFFFFu, C, key2, emit, s.wtmp);
    } else {
        ordered_threshold_compact_aux(C, T, budget, key2, emit, s.wtmp);
    }
    CUDDLE_STAMP(6, 4);
    if (tid == 0) sel_count[q] = min(min(C, rerank_k), sel_stride);
}

// Stage entry point: every candidate's position and estimate (bit pattern) in scan order.
__global__ void __launch_bounds__(256)
scan_all_rabitq_kernel(IndexDev ix, const float *rq_all, const float *qnorm_all, const uint32_t *leaves,
                       uint32_t lstride, const uint32_t *lcount, uint32_t lcap, uint64_t *out_pos, uint32_t *out_key,
                       uint32_t cand_stride, uint32_t *cand_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t D = ix.dim, W32 = ix.words32;
    RabitqLds r = rabitq_lds_carve(smem, D, lcap, 0);
    const ScanLds &s = r.s;
    const uint32_t nl = min(lcount[q], lcap);
    const float qnorm = qnorm_all[q];
    rabitq_stage_query(rq_all + (uint64_t)q * D, D, qnorm, r.rq, r.qsign);
    const uint32_t C = scan_load_leaves(ix, s, leaves + (uint64_t)q * lstride, nl, 0, nullptr);
    if (tid == 0) cand_count[q] = C;
    for (uint32_t sl = w; sl < nl; sl += nw) {
        const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t pos = s.lbeg[sl] + i;
            float e = rabitq_estimate(ix.rq_codes + (uint64_t)s.lbeg[sl] * W32, n, i, D, W32, r.rq, r.qsign, qnorm,
                                      ix.rq_norm[pos], ix.rq_ipoo[pos]);
            if (base + i < cand_stride) {
                out_pos[(uint64_t)q * cand_stride + base + i] = pos;
                out_key[(uint64_t)q * cand_stride + base + i] = __float_as_uint(e);
            }
        }
    }
}

// Reorders a query's selection (scan order) into phase-1 order (key, scan order) in place, so the
// final top-k can break distance ties by position in the selection. One block per query;
// dynamic LDS = next_pow2(max selection) * 8 bytes (the sort keys only: 128 KiB at the largest
// selection of 16384); the (position, key) pairs are permuted through the query's row of `pairs`
// ([nq][sel_stride] u64 of global scratch), since a gather in place would overwrite its own sources.
__global__ void __launch_bounds__(256)
selection_sort_f32_kernel(uint32_t *sel_pos, uint32_t *sel_key, uint32_t sel_stride, const uint32_t *sel_count,
                          uint64_t *pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = min(sel_count[q], sel_stride);
    const uint32_t npad = next_pow2(max(n, 1u));
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    uint64_t *pr = pairs + (uint64_t)q * sel_stride;
    for (uint32_t i = tid; i < npad; i += blockDim.x)
        skeys[i] = i < n ? (((uint64_t)f32_key(__uint_as_float(sk[i])) << 32) | i) : ~0ull;
    __syncthreads();
    block_bitonic_sort_u64(skeys, npad);
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint32_t j = (uint32_t)skeys[i];
        pr[i] = ((uint64_t)sk[j] << 32) | sp[j];
    }
    __syncthreads(); // every source has been read before the first destination is written
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint64_t v = pr[i]; // written by this same thread above
        sp[i] = (uint32_t)v;
        sk[i] = (uint32_t)(v >> 32);
    }
}

} // namespace cuddle
