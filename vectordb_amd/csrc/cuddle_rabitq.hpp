// cuddle_rabitq.hpp — RaBitQ phase 1 (reference: src/quantization/rabitq.rs:191-201, 298-363,
// 398-436; driver src/index/hierarchical_with_rabitq.rs:575-633).
//
// All sums run in the reference's order (strictly sequential f32 adds of separately rounded
// products), so estimates are bit-identical to the reference arithmetic.
#pragma once

#include <type_traits>

#include "cuddle_route.hpp"

namespace cuddle {

// rot[j][i] = rotation_matrix_t[i][j]: the layout in which consecutive lanes (consecutive output
// rows i) read consecutive addresses while every lane walks j in order.
__global__ void transpose_square_kernel(const float *src, float *dst, uint32_t d) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < d && j < d) dst[(uint64_t)j * d + i] = src[(uint64_t)i * d + j];
}

constexpr int kPrepQ = 8;       // queries per block of rabitq_query_prep_kernel
constexpr int kPrepRows = 256;  // output rows per block

// Query half of estimate_distances_batch_fast (rabitq.rs:398-420): qnorm = sqrt(sum q_j^2)
// (sequential), qn = q / qnorm, rq[i] = sum_j rot_t[i][j] * qn[j] (rotate_query, :191-201,
// sequential over j). A block handles kPrepQ queries and kPrepRows output rows: each rotation element is loaded once
// for several queries (their component j side by side in LDS: two 16-byte reads per j), thread t owns output row
// blockIdx.y * kPrepRows + t and walks j in order with AHEAD of its (coalesced) rotation loads in flight — one
// load per trip to memory made a lone 1536-d query wait 380 us for its rotation (64 in flight: 60 us), and the rows
// split over blocks are what a small batch has for parallelism; a batch that fills the GPU does better with 32 and the
// registers that saves (4096 x 768-d: 0.178 ms against 0.216). Every block of a query group computes the group's norms (a sequential sum
// from LDS); blockIdx.y == 0 writes them. rq is left untouched for qnorm < 1e-10.
// grid = (ceil(nq / kPrepQ), ceil(dim / kPrepRows)), block kPrepRows, dynamic LDS: kPrepQ * dim floats.
template <int AHEAD>
__global__ void __launch_bounds__(kPrepRows)
rabitq_query_prep_kernel(const float *queries, const float *rot, uint32_t nq, uint32_t dim, float *rq, float *qnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qn = reinterpret_cast<float *>(smem); // [dim][kPrepQ]
    __shared__ float s_norm[kPrepQ];
    const uint32_t q0 = blockIdx.x * kPrepQ;
    for (uint32_t t = threadIdx.x; t < kPrepQ * dim; t += blockDim.x) {
        const uint32_t qq = t / dim, j = t % dim, q = q0 + qq;
        qn[j * kPrepQ + qq] = q < nq ? queries[(uint64_t)q * dim + j] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < kPrepQ) {
        const uint32_t q = q0 + threadIdx.x;
        float s = 0.f;
        for (uint32_t j = 0; j < dim; ++j) {
            const float x = qn[j * kPrepQ + threadIdx.x];
            s += x * x;
        }
        const float nrm = sqrtf(s);
        s_norm[threadIdx.x] = nrm;
        if (q < nq && blockIdx.y == 0) qnorm[q] = nrm;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kPrepQ * dim; t += blockDim.x) {
        const float nrm = s_norm[t % kPrepQ];
        qn[t] = nrm >= 1e-10f ? qn[t] / nrm : 0.f;
    }
    __syncthreads();
    const uint32_t i = blockIdx.y * kPrepRows + threadIdx.x;
    if (i >= dim) return; // (no barrier below)
    float acc[kPrepQ];
#pragma unroll
    for (int qq = 0; qq < kPrepQ; ++qq) acc[qq] = 0.f;
    const float *col = rot + i;
    uint32_t j = 0;
    for (; j + AHEAD <= dim; j += AHEAD) {
        float r[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) r[u] = col[(uint64_t)(j + u) * dim];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
#pragma unroll
            for (int qq = 0; qq < kPrepQ; ++qq) acc[qq] += r[u] * qn[(j + u) * kPrepQ + qq];
        }
    }
    for (; j < dim; ++j) {
        const float r = col[(uint64_t)j * dim];
#pragma unroll
        for (int qq = 0; qq < kPrepQ; ++qq) acc[qq] += r * qn[j * kPrepQ + qq];
    }
#pragma unroll
    for (int qq = 0; qq < kPrepQ; ++qq)
        if (q0 + qq < nq && s_norm[qq] >= 1e-10f) rq[(uint64_t)(q0 + qq) * dim + i] = acc[qq];
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
    const uint32_t nfull = dim >> 5; // words whose 32 bits all belong to the vector
    uint32_t w = 0;
    // straight-line chunks (no condition between a load and its use, or the compiler sinks every load next to
    // its use and waits for each in turn): eight code words in flight per lane
    for (; w + 8 <= nfull; w += 8) {
        uint32_t cw[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) cw[u] = codes[(uint64_t)(w + u) * n + i];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t sg = ls[w + u] ^ ~cw[u];
#pragma unroll
            for (uint32_t b = 0; b < 32; ++b)
                sum += __uint_as_float(((sg << (31 - b)) & 0x80000000u) | __float_as_uint(la[32 * (w + u) + b]));
        }
    }
    for (; w < words32; ++w) {
        const uint32_t sg = ls[w] ^ ~codes[(uint64_t)w * n + i];
        const uint32_t lim = min(32u, dim - 32u * w);
        for (uint32_t b = 0; b < lim; ++b)
            sum += __uint_as_float(((sg << (31 - b)) & 0x80000000u) | __float_as_uint(la[32 * w + b]));
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

// ---------------------------------------------------------------------------------------------
// phase 1 in two steps: a cheap bound for every candidate, the exact estimate for the few that matter
// ---------------------------------------------------------------------------------------------
// The exact estimate costs one dependent f32 add per code bit (the reference's sequential sum), and only
// about one candidate in ten is selected. So every candidate first gets an approximate estimate from
// integer arithmetic with a proven error bound, and the exact sum is evaluated only where the bounds
// leave a choice:
//   * the query's magnitudes |rq_i| are quantised to kRqPlanes bits, |rq_i| ~ delta * v_i, and kept as bit
//     planes; with d_i = 1 where the code bit and the sign of rq_i make term i negative,
//       ip~ = delta * (sum_i v_i - 2 sum_p 2^p popcount(d & plane_p))            (exact integers),
//     and |ip~ - ip| <= B_q = sum_i ||rq_i| - delta v_i| + the rounding of the reference's f32 sum;
//   * est~ = (2 - 2 ip~ / ip_oo) norm qnorm, |est~ - est| <= E_c = norm_c (a_q + b_q / ip_oo_c);
//   * tau = the rerank_k-th smallest upper bound est~ + E: at least rerank_k candidates have their exact
//     estimate <= tau, so a candidate whose lower bound est~ - E is > tau cannot be selected;
//   * the others (the selection and a band around its boundary) get the exact sequential-order
//     estimate, two candidates per lane in packed f32 adds, and the selection runs on those values:
//     the same set, order and keys as evaluating every candidate exactly.
// Bits of the quantised magnitudes. The scan's cost grows with every plane (two operations per code word and
// plane); the error bound halves with every plane, and below ten the band of candidates that need the exact
// estimate starts to cost more than the planes save (10M x 768-d, k 100, rerank factor 10: 12 planes 2.03 ms,
// 11 1.96, 10 1.91, 9 1.90).
#ifndef CUDDLE_RQ_PLANES
#define CUDDLE_RQ_PLANES 10
#endif
constexpr uint32_t kRqPlanes = CUDDLE_RQ_PLANES;
constexpr uint32_t kRqListCap = 4096; // candidates whose exact estimate is needed; more: every candidate is evaluated exactly

inline size_t rabitq_scan_lds_bytes(uint32_t dim, uint32_t lcap, uint32_t tcap, uint32_t dist_cap) {
    return scan_lds_bytes(0, 0, lcap, tcap, 0) + 256 * 4 + (size_t)((dim + 3) & ~3u) * 4 + (size_t)((dim + 63) / 64) * 8 +
           (size_t)kRqListCap * 4 + 32 * 4 + (size_t)dist_cap * 4 + 96;
}

struct RabitqLds {
    ScanLds s;
    uint32_t *rhist; // [256] radix-select histogram
    float *rq;       // [dim]  |rotated query|
    uint32_t *qsign; // [2 * ceil(dim / 64)] sign bits of the rotated query
    uint32_t *list;  // [kRqListCap] scan positions of the candidates to evaluate exactly
    float *fsc;      // [16] unused; [16..31] the same memory seen as u32 counters
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
    r.list = p; p += kRqListCap;
    r.fsc = reinterpret_cast<float *>(p); p += 32;
    p = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
    r.est = reinterpret_cast<float *>(p);
    return r;
}

// Bit 31 of x, bits 0..30 of mag (one v_bfi_b32; the mask has to sit in a scalar register).
__device__ __forceinline__ float rabitq_signed(uint32_t x, float mag) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x7FFFFFFFu), "v"(__float_as_uint(mag)), "v"(x));
    return __uint_as_float(r);
}

// compute_inner_product_fast (rabitq.rs:298-321) for TWO vectors at once: the sequential sums of both run
// in the two halves of packed f32 adds (each half the same IEEE addition as the scalar sum), the sign of
// a term comes from bit 31 of the bit-reversed (query sign ^ ~code) word, which a 64-bit shift moves on
// for both vectors in one instruction. ca / cb: the vectors' code columns (word w at ca[w * na]).
__device__ __forceinline__ void rabitq_ip_pair(const uint32_t *ca, uint32_t na, const uint32_t *cb, uint32_t nb, uint32_t dim,
                                               uint32_t words32, const LdsF32 *la, const LdsU32 *ls, float &ipa, float &ipb) {
    v2f sum = {0.f, 0.f};
    const uint32_t nfull = dim >> 5;
    uint32_t w = 0;
    auto word = [&](uint32_t w, uint32_t wa, uint32_t wb, uint32_t lim) {
        const uint32_t q = ls[w];
        uint64_t y = ((uint64_t)__builtin_bitreverse32(q ^ ~wb) << 32) | __builtin_bitreverse32(q ^ ~wa);
        if (lim == 32) {
#pragma unroll
            for (uint32_t b = 0; b < 32; ++b) {
                const float m = la[32 * w + b];
                const v2f t = {rabitq_signed((uint32_t)y, m), rabitq_signed((uint32_t)(y >> 32), m)};
                sum += t;
                y <<= 1;
            }
        } else {
            for (uint32_t b = 0; b < lim; ++b) {
                const float m = la[32 * w + b];
                const v2f t = {rabitq_signed((uint32_t)y, m), rabitq_signed((uint32_t)(y >> 32), m)};
                sum += t;
                y <<= 1;
            }
        }
    };
    for (; w + 8 <= nfull; w += 8) { // straight-line chunks: sixteen loads in flight, no condition before their use
        uint32_t wa[8], wb[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            wa[u] = ca[(uint64_t)(w + u) * na];
            wb[u] = cb[(uint64_t)(w + u) * nb];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) word(w + u, wa[u], wb[u], 32);
    }
    for (; w < words32; ++w) word(w, ca[(uint64_t)w * na], cb[(uint64_t)w * nb], min(32u, dim - 32u * w));
    ipa = sum.x;
    ipb = sum.y;
}

// estimate_distance_fast (rabitq.rs:335-363) given the inner product.
__device__ __forceinline__ float rabitq_estimate_from_ip(float ip, float qnorm, float norm, float ipoo) {
    if (qnorm < 1e-10f) return norm * norm;
    if (norm < 1e-10f) return qnorm * qnorm;
    const float est_ip = ip / ipoo;
    const float dn = 2.0f - 2.0f * est_ip;
    return dn * norm * qnorm;
}

// The bound's query side lives in global memory, one row of rq_plane_stride(words32) u32 per query, written once
// per batch by rabitq_query_planes_kernel and read by the scan through scalar loads (the row is the same for every
// lane, so its words sit in scalar registers and cost the vector unit nothing):
//   [0] delta (f32)  [1] a_q (f32)  [2] b_q (f32)   (E_c = norm_c * (a_q + b_q / ip_oo_c))   [3] sum of the quantised values
//   [4] sum of the quantised values with the signs of rq (i32)
//   [8 + 12 w + 0] sign bits of rq (dimensions 32 w .. 32 w + 31)   [8 + 12 w + 1 + p] bit plane p of the quantised magnitudes
constexpr uint32_t kRqWordStride = (kRqPlanes + 1 + 3) & ~3u; // sign word + planes, padded to whole 16-byte pieces
__host__ __device__ inline uint32_t rq_plane_stride(uint32_t words32) { return 8 + kRqWordStride * words32; }

// Quantises |rq| of every query to kRqPlanes bits and writes its row (layout above). One wave per query.
// pieces != null: also the signed quantised values as two 5-bit pieces, [nq][2][dim rounded up to 64] i8
// (value = 32 * pieces[1] + pieces[0], zero past dim), the operand of rq_bounds_grouped_kernel.
__global__ void __launch_bounds__(256)
rabitq_query_planes_kernel(const float *rq_all, const float *qnorm_all, uint32_t nq, uint32_t dim, uint32_t words32,
                           uint32_t *rows, int8_t *pieces) {
    const uint32_t lane = threadIdx.x & 63, q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float *rq = rq_all + (uint64_t)q * dim;
    uint32_t *row = rows + (uint64_t)q * rq_plane_stride(words32);
    const float qnorm = qnorm_all[q];
    const bool live = qnorm >= 1e-10f; // rq is undefined for a zero query (the scan does not use the bounds then)
    const uint32_t dpad = (dim + 63) & ~63u;
    float mx = 0.f;
    for (uint32_t i = lane; i < dim; i += 64) mx = fmaxf(mx, live ? fabsf(rq[i]) : 0.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    constexpr uint32_t VMAX = (1u << kRqPlanes) - 1;
    const float delta = mx / (float)VMAX, inv = mx > 0.f ? (float)VMAX / mx : 0.f;
    float res = 0.f, asum = 0.f;
    uint32_t vsum = 0;
    int ssum = 0;
    static_assert(kRqPlanes <= 12, "the signed pieces are bytes: low 5 bits, the rest (at most 7 bits)");
    for (uint32_t i = lane; i < dpad; i += 64) {
        const uint32_t bits = (i < dim && live) ? __float_as_uint(rq[i]) : 0u;
        const float a = __uint_as_float(bits & 0x7FFFFFFFu);
        uint32_t v = (uint32_t)rintf(a * inv);
        v = (a == a && v <= VMAX) ? v : (a == a ? VMAX : 0u);
        res += fabsf(a - delta * (float)v);
        asum += a;
        vsum += v;
        const int sv = (bits >> 31) ? -(int)v : (int)v;
        ssum += sv;
        if (pieces) {
            int8_t *pc = pieces + (uint64_t)q * 2 * dpad;
            pc[i] = (int8_t)((bits >> 31) ? -(int)(v & 31u) : (int)(v & 31u));
            pc[dpad + i] = (int8_t)((bits >> 31) ? -(int)(v >> 5) : (int)(v >> 5));
        }
        const uint64_t sb = __ballot((bits >> 31) != 0);
        uint32_t *w0 = row + 8 + (size_t)kRqWordStride * (i >> 5); // i is a multiple of 64 in lane 0
        const bool second = (i >> 5) + 1 < words32;
        if (lane == 0) {
            w0[0] = (uint32_t)sb;
            if (second) w0[kRqWordStride] = (uint32_t)(sb >> 32);
        }
#pragma unroll
        for (uint32_t p = 0; p < kRqPlanes; ++p) {
            const uint64_t bal = __ballot((v >> p) & 1u);
            if (lane == 0) {
                w0[1 + p] = (uint32_t)bal;
                if (second) w0[kRqWordStride + 1 + p] = (uint32_t)(bal >> 32);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        res += __shfl_xor(res, o, 64);
        asum += __shfl_xor(asum, o, 64);
        vsum += __shfl_xor(vsum, o, 64);
        ssum += __shfl_xor(ssum, o, 64);
    }
    if (lane == 0) {
        row[4] = (uint32_t)ssum;
        // |ip~ - ip| <= B: quantisation residue (computed in f32: 0.2 % slack) + the rounding of the reference's
        // sequential f32 sum and of delta * integer (each below dim * 2^-23 times the sum of magnitudes)
        const float B = 1.002f * res + ((float)dim + 8.0f) * 1.2e-7f * (asum + res);
        // E_c = norm_c qnorm (2 B / ip_oo + 1e-5 (1 + (A + B) / ip_oo)), 0.1 % slack: the second term covers the
        // roundings of (2 - 2 ip / ip_oo) norm qnorm in either evaluation and of the bound itself
        row[0] = __float_as_uint(delta);
        row[1] = __float_as_uint(1.001f * qnorm * 1e-5f);
        row[2] = __float_as_uint(1.001f * qnorm * (2.0f * B + 1e-5f * (asum + B)));
        row[3] = vsum;
    }
}

// Upper (returned) and lower bound of the estimate from the integer S = sum_i (+-1) sgn(rq_i) v_i; both rounded
// outwards. The one formula of the per-query and the leaf-by-leaf kernel (their bounds agree bit for bit).
// 1 / ip_oo is the hardware's reciprocal (1 ulp): against the reference's division the quotient moves by under
// 3e-7 of itself, the estimate by under 1e-6 norm qnorm - inside the 1e-5 norm qnorm that a_q sets aside for the
// roundings of either evaluation - and b_q / ip_oo by a relative 3e-7, inside the 0.1 % slack of b_q.
__device__ __forceinline__ float rabitq_bounds_from_sum(int S, float delta, float aq, float bq, float qnorm, float norm, float ipoo,
                                                        float &lb) {
    if (norm < 1e-10f) { // exact (rabitq.rs:342-349)
        lb = qnorm * qnorm;
        return lb;
    }
    const float inv = __builtin_amdgcn_rcpf(ipoo);
    const float ip = delta * (float)S;
    const float est = (2.0f - 2.0f * (ip * inv)) * norm * qnorm;
    const float E = norm * (aq + bq * inv), round_out = 1e-6f * fabsf(est) + 1e-6f * E;
    lb = est - E - round_out;
    return est + E + round_out;
}

// Upper bound est~ + E (returned) and lower bound est~ - E (lb) of the estimate of one vector; see the section
// comment. Both are rounded outwards. Code word w of the vector is codes[off + w * n] (leaf blocks are word-major
// [w][n]). `row` (the query's row of rabitq_query_planes_kernel) is the same for every lane of the wave: the planes
// come through scalar loads. UNIFORM: `codes` and `n` are wave-uniform too (the wave's lanes are consecutive vectors
// of one leaf, off = the lane's place in it), so the word's address is a scalar row base plus the lane's own bytes;
// otherwise every lane brings its own leaf (off = where the vector's word 0 lies in the whole code array).
template <bool UNIFORM>
__device__ __forceinline__ float rabitq_bounds(const uint32_t *__restrict__ codes, uint32_t n, uint32_t off, uint32_t words32,
                                               const uint32_t *__restrict__ row, float qnorm, float norm, float ipoo, float &lb) {
    if (norm < 1e-10f) { // exact (rabitq.rs:342-349)
        lb = qnorm * qnorm;
        return lb;
    }
    uint32_t cnt[kRqPlanes];
#pragma unroll
    for (uint32_t p = 0; p < kRqPlanes; ++p) cnt[p] = 0;
    // straight-line chunks of 8 and 1 code words (no condition between a load and its use, or the compiler sinks
    // every load next to its use and waits for each in turn)
    auto chunk = [&](uint32_t w, auto ch) {
        constexpr uint32_t CH = decltype(ch)::value;
        uint32_t cw[CH];
#pragma unroll
        for (uint32_t u = 0; u < CH; ++u) {
            if (UNIFORM) {
                cw[u] = (codes + (uint64_t)(w + u) * n)[off];
            } else {
                cw[u] = codes[off];
                off += n;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < CH; ++u) {
            const uint32_t *pw = row + 8 + kRqWordStride * (w + u);
            const uint32_t d = pw[0] ^ ~cw[u]; // bit i set: term i is negative
#pragma unroll
            for (uint32_t p = 0; p < kRqPlanes; ++p) popc_acc(cnt[p], d & pw[1 + p]);
        }
    };
    uint32_t w = 0;
    for (; w + 8 <= words32; w += 8) chunk(w, std::integral_constant<uint32_t, 8>());
    for (; w < words32; ++w) chunk(w, std::integral_constant<uint32_t, 1>());
    uint32_t K = 0;
#pragma unroll
    for (uint32_t p = 0; p < kRqPlanes; ++p) K += cnt[p] << p;
    return rabitq_bounds_from_sum((int)row[3] - 2 * (int)K, __uint_as_float(row[0]), __uint_as_float(row[1]), __uint_as_float(row[2]),
                                  qnorm, norm, ipoo, lb);
}

// compute_inner_product_fast for two vectors (see rabitq_ip_pair) whose codes are read from the vector-major copy:
// ra / rb = the vectors' rows of `vm_words` u32 (a multiple of 4, zero padded), 16-byte aligned.
__device__ __forceinline__ void rabitq_ip_pair_rows(const uint4 *ra, const uint4 *rb, uint32_t dim, uint32_t words32,
                                                    const LdsF32 *la, const LdsU32 *ls, float &ipa, float &ipb) {
    v2f sum = {0.f, 0.f};
    auto word = [&](uint32_t w, uint32_t wa, uint32_t wb) {
        const uint32_t q = ls[w];
        uint64_t y = ((uint64_t)__builtin_bitreverse32(q ^ ~wb) << 32) | __builtin_bitreverse32(q ^ ~wa);
        if (32 * w + 32 <= dim) {
#pragma unroll
            for (uint32_t b = 0; b < 32; ++b) {
                const float m = la[32 * w + b];
                const v2f t = {rabitq_signed((uint32_t)y, m), rabitq_signed((uint32_t)(y >> 32), m)};
                sum += t;
                y <<= 1;
            }
        } else {
            const uint32_t lim = dim > 32 * w ? dim - 32 * w : 0;
            for (uint32_t b = 0; b < lim; ++b) {
                const float m = la[32 * w + b];
                const v2f t = {rabitq_signed((uint32_t)y, m), rabitq_signed((uint32_t)(y >> 32), m)};
                sum += t;
                y <<= 1;
            }
        }
    };
    uint32_t w = 0;
    for (; w + 8 <= words32; w += 8) { // four 16-byte loads in flight, no condition before their use
        const uint4 a0 = ra[w / 4], a1 = ra[w / 4 + 1], b0 = rb[w / 4], b1 = rb[w / 4 + 1];
        word(w, a0.x, b0.x); word(w + 1, a0.y, b0.y); word(w + 2, a0.z, b0.z); word(w + 3, a0.w, b0.w);
        word(w + 4, a1.x, b1.x); word(w + 5, a1.y, b1.y); word(w + 6, a1.z, b1.z); word(w + 7, a1.w, b1.w);
    }
    for (; w < words32; w += 4) { // the padding words of the last piece cover no dimension: word() adds nothing for them
        const uint4 a0 = ra[w / 4], b0 = rb[w / 4];
        word(w, a0.x, b0.x); word(w + 1, a0.y, b0.y); word(w + 2, a0.z, b0.z); word(w + 3, a0.w, b0.w);
    }
    ipa = sum.x;
    ipb = sum.y;
}

// One block per query (block b takes the query scan_tables_kernel ranked b-th by work): estimated
// distance of the vectors of every reached leaf (compute_rabitq_distances / collect_leaf_candidates,
// hierarchical_with_rabitq.rs:581-619), then the rerank_k smallest by (estimate, scan position) — the
// first rerank_k of the reference's stable sort (select_top_candidates, :623-633) — emitted in scan
// order with the estimate's bit pattern as key. Bounds first, exact estimates where they leave a choice
// (section comment above); `exact_all` != 0 evaluates every candidate exactly (A/B and fallback path).
// Values live in LDS (or the global spill buffer when a query has more than dist_cap candidates).
// stats: [0] += candidates evaluated exactly, [1] += queries that took the exact-for-all path.
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8))) // two blocks of 16 waves per CU
scan_select_rabitq_kernel(IndexDev ix, const float *rq_all, const float *qnorm_all, const uint32_t *tables,
                          uint32_t tstride, uint32_t lcap, uint32_t tcap, uint32_t rerank_k,
                          uint32_t dist_cap, float *gest, uint32_t gest_stride, uint32_t *sel_pos, uint32_t *sel_key,
                          uint32_t sel_stride, uint32_t *sel_count, uint32_t *cand_count, uint32_t exact_all,
                          unsigned long long *stats, const uint32_t *__restrict__ qrows, uint32_t pre_bounds, uint32_t pack_ok) {
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
    float *ge = gest + (uint64_t)q * 2 * gest_stride; // the query's spill row: estimates | lower bounds
    float *glb = ge + gest_stride;
    auto est_at = [&](uint32_t i) -> float { return use_lds ? r.est[i] : ge[i]; };
    auto est_put = [&](uint32_t i, float v) { if (use_lds) r.est[i] = v; else ge[i] = v; };
    const bool all = C <= rerank_k;
    // the bounds pay when a selection has to be made and the query has a direction (a zero query's estimates are norm^2)
    bool filtered = !exact_all && !all && qnorm >= 1e-10f;
    uint32_t *isc = reinterpret_cast<uint32_t *>(r.fsc) + 16;
    CUDDLE_STAMP(6, 1);
    if (filtered) {
        if (tid == 0) isc[1] = 0; // members
        if (pre_bounds) {
            // rq_bounds_grouped_kernel has left the bounds in the query's spill row: the upper bounds move to where the
            // selection reads them (pre_bounds == 2, a debugging aid: they are recomputed below and compared instead)
            if (use_lds && pre_bounds == 1)
                for (uint32_t i = tid; i < C; i += blockDim.x) r.est[i] = ge[i];
        }
        // this query's row of quantised magnitudes: a wave-uniform address, so its words are read into scalar registers
        const uint32_t *row = qrows + (uint64_t)__builtin_amdgcn_readfirstlane(q) * rq_plane_stride(W32);
        // Tiles of 64 consecutive scan positions, whatever leaves they fall into (a leaf of 103 vectors cut by itself
        // would fill its second wave to 39 lanes of 64, and this loop is bound by the vector unit's issue rate). The
        // leaf slot of every tile's first position goes through the member list's storage, unused until tau is known.
        // (pack_ok == 0, a code array of 2^32 words or more: the tiles cut at leaf ends by scan_build_tiles instead.)
        const uint32_t nt = pack_ok ? (C + 63) >> 6 : ntiles;
        const bool slot_tab = pack_ok && nt <= kRqListCap;
        if (slot_tab && pre_bounds != 1) {
            for (uint32_t t = tid; t < nt; t += blockDim.x) r.list[t] = scan_slot_of(s.lscan, nl, 64 * t);
            __syncthreads();
        }
        for (uint32_t t = w; t < nt && pre_bounds != 1; t += nw) {
            uint32_t sl0, first;
            if (pack_ok) {
                first = 64 * t;
                sl0 = __builtin_amdgcn_readfirstlane(slot_tab ? r.list[t] : scan_slot_of(s.lscan, nl, first));
            } else {
                const uint32_t tile = __builtin_amdgcn_readfirstlane(s.tiles[t]);
                sl0 = tile >> 12;
                first = __builtin_amdgcn_readfirstlane(s.lscan[sl0]) + ((tile & 4095u) << 6);
            }
            const uint32_t end0 = __builtin_amdgcn_readfirstlane(s.lscan[sl0 + 1]);
            const uint32_t lim = min(pack_ok ? C : end0, first + 64); // the tile: scan positions [first, lim)
            const uint32_t i = first + lane, ic = min(i, lim - 1);
            float ub, lb;
            if (lim <= end0) {
                // the whole tile lies in one leaf: everything but the lane's own vector is the same for the wave
                const uint32_t base = __builtin_amdgcn_readfirstlane(s.lscan[sl0]), n = end0 - base;
                const uint32_t beg = __builtin_amdgcn_readfirstlane(s.lbeg[sl0]);
                const uint32_t k = ic - base, pos = beg + k;
                ub = rabitq_bounds<true>(ix.rq_codes + (uint64_t)beg * W32, n, k, W32, row, qnorm, ix.rq_norm[pos], ix.rq_ipoo[pos], lb);
            } else {
                uint32_t sl = sl0;
                while (ic >= s.lscan[sl + 1]) ++sl; // (skips empty leaves as well)
                const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base, beg = s.lbeg[sl];
                const uint32_t k = ic - base, pos = beg + k;
                ub = rabitq_bounds<false>(ix.rq_codes, n, beg * W32 + k, W32, row, qnorm, ix.rq_norm[pos], ix.rq_ipoo[pos], lb);
            }
            if (i < lim) {
                if (pre_bounds == 2 && stats &&
                    (__float_as_uint(ub) != __float_as_uint(ge[i]) || __float_as_uint(lb) != __float_as_uint(glb[i])))
                    atomicAdd(&stats[2], 1ull);
                est_put(i, ub);
                glb[i] = lb;
            }
        }
        __syncthreads();
        CUDDLE_STAMP(6, 2);
        // tau = the rerank_k-th smallest upper bound
        uint32_t bdummy;
        const uint32_t Tub = block_radix_select_aligned(C, rerank_k, [&](uint32_t i) { return f32_key(est_at(i)); }, r.rhist, s.misc,
                                                        s.wtmp, bdummy);
        const float tau = f32_from_key(Tub);
        // members: lower bound <= tau (written so that a NaN anywhere keeps the candidate); the others can never be selected
        for (uint32_t i = tid; i < C; i += blockDim.x) {
            if (!(glb[i] > tau) || Tub == 0xFFFFFFFFu) {
                const uint32_t slot = atomicAdd(&isc[1], 1u);
                if (slot < kRqListCap) r.list[slot] = i;
            } else {
                est_put(i, __builtin_inff());
            }
        }
        __syncthreads();
        const uint32_t ns = isc[1];
        if (ns > kRqListCap) {
            filtered = false; // too many to list: every candidate exactly (below)
        } else {
            // exact estimates of the members, two per lane
            for (uint32_t j = 2 * tid; j < ns; j += 2 * blockDim.x) {
                const uint32_t ia = r.list[j], ib = r.list[min(j + 1, ns - 1)];
                const uint32_t sa = scan_slot_of(s.lscan, nl, ia), sb = scan_slot_of(s.lscan, nl, ib);
                const uint32_t ka = ia - s.lscan[sa], kb = ib - s.lscan[sb];
                const uint32_t pa = s.lbeg[sa] + ka, pb = s.lbeg[sb] + kb;
                float ipa, ipb;
                // the vector-major copy: one vector's code is 4 * vm_words contiguous bytes (the word-major leaf blocks
                // would cost one cache line per code word here)
                rabitq_ip_pair_rows(reinterpret_cast<const uint4 *>(ix.rq_codes_vm + (uint64_t)pa * ix.rq_vm_words),
                                    reinterpret_cast<const uint4 *>(ix.rq_codes_vm + (uint64_t)pb * ix.rq_vm_words), D, W32,
                                    (const LdsF32 *)r.rq, (const LdsU32 *)r.qsign, ipa, ipb);
                est_put(ia, rabitq_estimate_from_ip(ipa, qnorm, ix.rq_norm[pa], ix.rq_ipoo[pa]));
                if (j + 1 < ns) est_put(ib, rabitq_estimate_from_ip(ipb, qnorm, ix.rq_norm[pb], ix.rq_ipoo[pb]));
            }
            if (tid == 0 && stats) atomicAdd(&stats[0], (unsigned long long)ns);
        }
        __syncthreads();
    }
    if (!filtered) {
        for (uint32_t t = w; t < ntiles; t += nw) {
            const uint32_t tile = s.tiles[t], sl = tile >> 12, i = ((tile & 4095u) << 6) + lane;
            const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
            if (i < n) {
                const uint32_t pos = s.lbeg[sl] + i;
                est_put(base + i, rabitq_estimate(ix.rq_codes + (uint64_t)s.lbeg[sl] * W32, n, i, D, W32, r.rq, r.qsign, qnorm,
                                                  ix.rq_norm[pos], ix.rq_ipoo[pos]));
            }
        }
        if (tid == 0 && stats) {
            atomicAdd(&stats[0], (unsigned long long)C);
            atomicAdd(&stats[1], 1ull);
        }
        __syncthreads();
    }
    CUDDLE_STAMP(6, 3);
    auto key = [&](uint32_t i) -> uint32_t { return f32_key(est_at(i)); };
    auto key2 = [&](uint32_t i, uint32_t &) -> uint32_t { return key(i); };
    uint32_t T = 0xFFFFFFFFu, budget = 0;
    if (filtered) // every other candidate holds +inf: the threshold and its tie budget are those of the listed ones
        T = block_radix_select_aligned(isc[1], rerank_k, [&](uint32_t j) { return f32_key(est_at(r.list[j])); }, r.rhist, s.misc,
                                       s.wtmp, budget);
    else if (!all)
        T = block_radix_select_aligned(C, rerank_k, key, r.rhist, s.misc, s.wtmp, budget);
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    auto emit = [&](uint32_t i, uint32_t, uint32_t, uint32_t rank) {
        if (rank < sel_stride) {
            uint32_t sl = scan_slot_of(s.lscan, nl, i);
            sp[rank] = s.lbeg[sl] + (i - s.lscan[sl]);
            sk[rank] = __float_as_uint(est_at(i));
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
