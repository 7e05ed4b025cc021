// cuddle_kernels.hpp — device code of libcuddle_hip (gfx950 / CDNA4, wave64).
//
// Kernels of the batched two-phase query path. Every arithmetic routine states which
// reference routine it reproduces; floating-point results are bit-identical to that routine's
// accumulation order (fused multiply-add exactly where the reference uses vfmaq_f32), so the
// translation unit must be compiled with -ffp-contract=off.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cuddle {

constexpr int kWave = 64;
constexpr uint32_t kMetricL2 = 0, kMetricCosine = 1, kMetricDot = 2;
constexpr uint32_t kSortCap = 4096;      // max probes_per_level (sorted selection lives in LDS)
constexpr uint32_t kMaxRerankK = 16384;  // max phase-1 selection per query (final sort lives in LDS)

// Device view of an index (all pointers are device pointers).
struct IndexDev {
    uint32_t dim;
    uint32_t words;        // u64 words per binary code = ceil(dim/64)
    uint32_t metric;
    uint32_t code_kind;
    uint32_t n_nodes;
    uint32_t n_roots;
    uint64_t n_vectors;
    const float *centroids;       // [n_nodes][dim]
    const float *centroid_sqnorm; // [n_nodes]  fma-chain sum of squares (MFMA routing)
    const uint32_t *child_off;    // [n_nodes+1]
    const uint32_t *children;
    const uint32_t *roots;
    const uint32_t *level_pos;    // [n_nodes]   position of the node in its level's breadth-first node list
    const uint32_t *vec_off;      // [n_nodes+1] leaf-contiguous positions
    const uint32_t *code_off;     // [n_nodes]   start of the leaf's planar code block, 16-byte units
    const uint64_t *ids;          // [n_vectors]
    const float *vectors;         // [n_vectors][dim]
    const uint4 *codes;           // planar per-leaf binary codes (see relayout_codes_kernel)
    // RaBitQ
    const float *rot_t;           // [dim][dim]  rotation_matrix_t (row i = output component i)
    const float *rot;             // [dim][dim]  its transpose: rot[j][i] = rot_t[i][j]
    const uint32_t *rq_codes;     // planar per-leaf: word-major [words32][n] per leaf
    const float *rq_norm;         // [n_vectors]
    const float *rq_ipoo;         // [n_vectors]
    uint32_t words32;             // ceil(dim/32)
    uint32_t rq_vm_words;         // u32 per vector in rq_codes_vm: words32 rounded up to a multiple of 4
    const uint32_t *rq_codes_vm;  // [n_vectors][rq_vm_words] the same codes vector-major, zero padded (16-byte aligned rows)
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------

// Debug builds (-DCUDDLE_STAMP_KERNEL=<id>) record the 100 MHz clock at up to 8 points of every
// block of ONE kernel (1 Hamming scan, 2 grouped routing distances, 3 register selection,
// 4 routing finish, 5 register prefilter, 6 RaBitQ scan, 10 leaf-by-leaf scan, 11 rerank); cuddle_debug_stamps() reads them back.
#ifdef CUDDLE_STAMP_KERNEL
constexpr uint32_t kStampBlocks = 16384;
__device__ uint64_t g_dbg_stamps[kStampBlocks * 8];
#define CUDDLE_STAMP_AT(kernel_id, n, block)                                                         \
    do {                                                                                             \
        if ((kernel_id) == CUDDLE_STAMP_KERNEL && threadIdx.x == 0 && (block) < kStampBlocks)        \
            g_dbg_stamps[(block) * 8 + (n)] = wall_clock64();                                        \
    } while (0)
#else
#define CUDDLE_STAMP_AT(kernel_id, n, block) do { } while (0)
#endif
#define CUDDLE_STAMP(kernel_id, n) CUDDLE_STAMP_AT(kernel_id, n, blockIdx.x)

// Order-preserving map f32 -> u32 for "a < b" comparisons as the reference's partial_cmp sees
// them: -0.0 == +0.0, NaN after everything.
__device__ __forceinline__ uint32_t f32_key(float d) {
    if (d != d) return 0xFFFFFFFFu;
    uint32_t b = __float_as_uint(d);
    if ((b << 1) == 0u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Pointers known to address LDS (a generic pointer makes the compiler issue flat loads/stores).
typedef __attribute__((address_space(3))) float LdsF32;
typedef __attribute__((address_space(3))) uint32_t LdsU32;
typedef __attribute__((address_space(3))) uint16_t LdsU16;
typedef __attribute__((address_space(3))) uint8_t LdsU8;

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// Inclusive prefix sum across the 64 lanes of a wave (all 64 active), by DPP moves instead of the LDS
// crossbar: row shifts 1/2/4/8 scan each 16-lane row, row_bcast 15 adds row 0's total to row 1 and
// row 2's to row 3, row_bcast 31 adds the total of rows 0-1 to rows 2 and 3. Lanes without a source add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_step(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v = dpp_add_step<0x111, 0xF>(v); // row_shr:1
    v = dpp_add_step<0x112, 0xF>(v); // row_shr:2
    v = dpp_add_step<0x114, 0xF>(v); // row_shr:4
    v = dpp_add_step<0x118, 0xF>(v); // row_shr:8
    v = dpp_add_step<0x142, 0xA>(v); // row_bcast:15
    v = dpp_add_step<0x143, 0xC>(v); // row_bcast:31
    return v;
}

// Smallest u32 of the wave (all 64 lanes active), in every lane; same DPP sequence, identity ~0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_step_u32(uint32_t v) {
    return min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_step_u32<0x111, 0xF>(v);
    v = dpp_min_step_u32<0x112, 0xF>(v);
    v = dpp_min_step_u32<0x114, 0xF>(v);
    v = dpp_min_step_u32<0x118, 0xF>(v);
    v = dpp_min_step_u32<0x142, 0xA>(v);
    v = dpp_min_step_u32<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// In-place exclusive prefix sum of lds[0..n) by the first wave of the block; lds[n] receives
// the total. Callers barrier before and after.
__device__ __forceinline__ void wave0_exclusive_scan_lds(uint32_t *lds, uint32_t n) {
    if (threadIdx.x >= 64) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t per = (n + 63) / 64;
    const uint32_t b = lane * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (b + i < n) sum += lds[b + i];
    uint32_t incl = wave_inclusive_sum(sum);
    uint32_t run = incl - sum;
    for (uint32_t i = 0; i < per; ++i)
        if (b + i < n) {
            uint32_t v = lds[b + i];
            lds[b + i] = run;
            run += v;
        }
    if (lane == 63) lds[n] = incl;
}

// Bitonic sort of n (power of two) u64 keys in LDS, ascending. All threads of the block call.
__device__ __forceinline__ void block_bitonic_sort_u64(uint64_t *s, uint32_t n) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (n >> 1); t += blockDim.x) { // the t-th pair of this step: no idle threads
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                const uint64_t a = s[i], b = s[ixj];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
            __syncthreads();
        }
    }
}

// The same sort done by wave 0 alone (the other waves skip it): no block barrier between the
// steps, which is what a block-wide sort of a few hundred keys spends its time on. The caller
// puts a __syncthreads() before and after.
__device__ __forceinline__ void wave0_bitonic_sort_u64(uint64_t *s, uint32_t n) {
    if (threadIdx.x >= 64) return;
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (n >> 1); t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j; // t-th pair of this step
                const uint64_t a = s[i], b = s[ixj];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// Sort by rank: every thread counts, for each of its (at most two) keys, the keys below it and stores the key
// at that position. The keys are distinct except for the padding value ~0, which sits behind the real keys and
// stays where it is. n^2 / blockDim comparisons on broadcast LDS reads and two barriers, against the 45 dependent
// steps of a bitonic network at n = 512 (20 us when wave 0 runs it alone). n: a power of two, at most
// 2 * blockDim. All threads of the block call; the caller has synchronised after writing the keys.
__device__ __forceinline__ void block_rank_sort_u64(uint64_t *s, uint32_t n) {
    if (n < 2) return;
    const uint32_t i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
    const uint64_t k0 = i0 < n ? s[i0] : ~0ull, k1 = i1 < n ? s[i1] : ~0ull;
    uint32_t r0 = 0, r1 = 0, j = 0;
    for (; j + 16 <= n; j += 16) { // sixteen keys per step: the reads are issued together, their latency is paid once
        uint64_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            r0 += v[u] < k0 ? 1u : 0u;
            r1 += v[u] < k1 ? 1u : 0u;
        }
    }
    for (; j < n; ++j) {
        const uint64_t a = s[j];
        r0 += a < k0 ? 1u : 0u;
        r1 += a < k1 ? 1u : 0u;
    }
    __syncthreads(); // every key has been read
    if (k0 != ~0ull) s[r0] = k0;
    if (k1 != ~0ull) s[r1] = k1;
    __syncthreads();
}

// Sort of n (power of two) keys (distinct, padded at the end with ~0) by the whole block.
__device__ __forceinline__ void block_sort_u64(uint64_t *s, uint32_t n) {
    if (n <= 512 && n <= 2 * blockDim.x)
        block_rank_sort_u64(s, n);
    else
        block_bitonic_sort_u64(s, n);
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Ordered threshold compaction. Over items i in [0, C) with integer keys key(i), emit every item
// with key < T and the first `budget` items (in index order) with key == T, calling
// emit(i, key, aux, rank) with rank = the item's position among the emitted ones in index order
// and aux = whatever key(i, aux) stored for the item.
// Each wave owns one contiguous segment; wtmp needs 2 * (blockDim.x / 64) u32 of LDS.
// All threads of the block call; contains two barriers.
template <class KeyFn, class EmitFn>
__device__ __forceinline__ void ordered_threshold_compact_aux(uint32_t C, uint32_t T, uint32_t budget,
                                                          KeyFn key, EmitFn emit, uint32_t *wtmp) {
    const uint32_t nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t seg = (C + nw - 1) / nw;
    seg = (seg + 63) & ~63u;
    const uint32_t beg = min(w * seg, C), end = min(beg + seg, C);
    // Both passes read four 64-element steps at a time so that the key loads of a step do not wait
    // for the ballots of the one before. key(i, aux) may hand a per-element value on to emit.
    constexpr int kSteps = 4;
    uint32_t n_lt = 0, n_tie = 0;
    for (uint32_t i0 = beg; i0 < end; i0 += 64 * kSteps) {
        uint32_t kv[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const uint32_t i = i0 + 64 * u + lane;
            uint32_t aux;
            kv[u] = (i < end) ? key(i, aux) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const bool in = i0 + 64 * u + lane < end;
            n_lt += __popcll(__ballot(in && kv[u] < T));
            n_tie += __popcll(__ballot(in && kv[u] == T));
        }
    }
    if (lane == 0) {
        wtmp[w] = n_lt;
        wtmp[nw + w] = n_tie;
    }
    __syncthreads();
    uint32_t lt_before = 0, tie_before = 0;
    for (uint32_t v = 0; v < w; ++v) {
        lt_before += wtmp[v];
        tie_before += wtmp[nw + v];
    }
    uint32_t out_before = lt_before + min(tie_before, budget);
    const uint64_t lm = lanemask_lt();
    for (uint32_t i0 = beg; i0 < end; i0 += 64 * kSteps) {
        uint32_t kv[kSteps], aux[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const uint32_t i = i0 + 64 * u + lane;
            aux[u] = 0;
            kv[u] = (i < end) ? key(i, aux[u]) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const uint32_t i = i0 + 64 * u + lane;
            const bool in = i < end;
            const bool tie = in && kv[u] == T;
            const uint64_t tb = __ballot(tie);
            const uint32_t tie_rank = tie_before + __popcll(tb & lm);
            const bool flag = in && (kv[u] < T || (tie && tie_rank < budget));
            const uint64_t fb = __ballot(flag);
            if (flag) emit(i, kv[u], aux[u], out_before + (uint32_t)__popcll(fb & lm));
            tie_before += __popcll(tb);
            out_before += __popcll(fb);
        }
    }
    __syncthreads();
}

// The same with key(i) and emit(i, key, rank).
template <class KeyFn, class EmitFn>
__device__ __forceinline__ void ordered_threshold_compact(uint32_t C, uint32_t T, uint32_t budget, KeyFn key, EmitFn emit,
                                                          uint32_t *wtmp) {
    ordered_threshold_compact_aux(
        C, T, budget, [&](uint32_t i, uint32_t &) -> uint32_t { return key(i); },
        [&](uint32_t i, uint32_t kv, uint32_t, uint32_t rank) { emit(i, kv, rank); }, wtmp);
}

// Ordered threshold selection over packed small keys (DistT = u8 or u16, `dist` 4-byte aligned and
// readable up to the next multiple of 4 bytes): the same result as ordered_threshold_compact, but
// every lane walks its own contiguous run of 32-bit words, so the loop body has no cross-lane
// operation: pass 1 counts keys < T and == T per lane, one wave scan and a per-wave table turn the
// counts into each lane's starting ranks, pass 2 calls emit(i, rank) for the selected items.
// The words of a lane are an odd number apart from the next lane's (no LDS bank conflicts).
// Keys are compared a word at a time: for fields of B bits and n <= 2^(B-1),
// ((L + n * ONES) - (x & L)) & ~x & H has a field's top bit set exactly when the field is < n
// (H = the top bit of every field, L = ~H; neither the sum nor the difference crosses a field).
// When T + 1 > 2^(B-1) the fields are compared one by one instead.
// wtmp needs 2 * (blockDim.x / 64) u32. All threads call; contains two barriers.
// WordPtr: pointer to the keys seen as u32 words, in the address space they live in.
template <class DistT, class WordPtr, class EmitFn>
__device__ __forceinline__ void chunked_threshold_select(WordPtr words, uint32_t C, uint32_t T, uint32_t budget,
                                                         EmitFn emit, uint32_t *wtmp) {
    constexpr uint32_t EPW = 4 / sizeof(DistT), BITS = 8 * sizeof(DistT), MASK = (1u << BITS) - 1;
    constexpr uint32_t H = BITS == 8 ? 0x80808080u : 0x80008000u, L = ~H, ONES = BITS == 8 ? 0x01010101u : 0x00010001u;
    const uint32_t nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nwords = (C + EPW - 1) / EPW;
    const uint32_t mw = (((nwords + nw - 1) / nw + 63) / 64) | 1u; // words per lane, odd
    const uint32_t wb = min((w * 64 + lane) * mw, nwords), we = min(wb + mw, nwords);
    // fields past the last key read as all ones: never < T, never == T (T <= dim + 1 < all ones)
    const uint32_t tail_valid = C - (nwords - 1) * EPW;
    const uint32_t tail_or = tail_valid < EPW ? ~0u << (tail_valid * BITS) : 0u;
    auto load = [&](uint32_t j) -> uint32_t {
        if (j >= we) return 0xFFFFFFFFu;
        const uint32_t x = words[j];
        return j == nwords - 1 ? x | tail_or : x;
    };
    const bool swar = T + 1 <= (1u << (BITS - 1));
    const uint32_t a_lt = L + T * ONES, a_le = L + (T + 1) * ONES; // used when swar
    uint32_t n_lt = 0, n_tie = 0;
    for (uint32_t j0 = wb; j0 < we; j0 += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load(j0 + u);
        if (swar) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t lo = v[u] & L, nx = ~v[u] & H;
                const uint32_t lt = (a_lt - lo) & nx, le = (a_le - lo) & nx;
                n_lt += __popc(lt);
                n_tie += __popc(le ^ lt);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (uint32_t e = 0; e < EPW; ++e) {
                    const uint32_t kv = (v[u] >> (e * BITS)) & MASK;
                    n_lt += kv < T ? 1u : 0u;
                    n_tie += kv == T ? 1u : 0u;
                }
            }
        }
    }
    CUDDLE_STAMP(1, 5);
    const uint32_t incl_lt = wave_inclusive_sum(n_lt), incl_tie = wave_inclusive_sum(n_tie);
    if (lane == 63) {
        wtmp[w] = incl_lt;
        wtmp[nw + w] = incl_tie;
    }
    __syncthreads();
    CUDDLE_STAMP(1, 6);
    uint32_t lt_run = incl_lt - n_lt, tie_run = incl_tie - n_tie;
    for (uint32_t v = 0; v < w; ++v) {
        lt_run += wtmp[v];
        tie_run += wtmp[nw + v];
    }
    if (swar) {
        // The selected keys are few, so a lane first folds the outcome of a whole group of its words into two
        // bit masks (one bit per key: below the threshold / equal to it) and then walks the set bits of the
        // group once: the walk is as long as the busiest lane's selection in the group, not the sum over the
        // words of the busiest lane per word.
        constexpr uint32_t G = 32 / EPW; // words per group: 32 keys
        for (uint32_t j0 = wb; j0 < we; j0 += G) {
            uint32_t v[G];
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) v[u] = load(j0 + u);
            uint32_t mlt = 0, mtie = 0;
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {
                const uint32_t lo = v[u] & L, nx = ~v[u] & H;
                const uint32_t lt = (a_lt - lo) & nx, tie = ((a_le - lo) & nx) ^ lt;
                // the top bit of every field -> one bit per field
                uint32_t clt, ctie;
                if (BITS == 8) {
                    clt = ((lt >> 7) & 1u) | ((lt >> 14) & 2u) | ((lt >> 21) & 4u) | ((lt >> 28) & 8u);
                    ctie = ((tie >> 7) & 1u) | ((tie >> 14) & 2u) | ((tie >> 21) & 4u) | ((tie >> 28) & 8u);
                } else {
                    clt = ((lt >> 15) & 1u) | ((lt >> 30) & 2u);
                    ctie = ((tie >> 15) & 1u) | ((tie >> 30) & 2u);
                }
                mlt |= clt << (u * EPW);
                mtie |= ctie << (u * EPW);
            }
            // ties stop mattering to this lane once the budget is used up before it
            uint32_t m = mlt | (tie_run < budget ? mtie : 0u);
            while (m) {
                const uint32_t bit = __ffs(m) - 1, i = j0 * EPW + bit;
                if ((mlt >> bit) & 1u) {
                    emit(i, lt_run + min(tie_run, budget));
                    ++lt_run;
                } else {
                    if (tie_run < budget) emit(i, lt_run + tie_run);
                    ++tie_run;
                }
                m &= m - 1;
            }
        }
    } else {
        for (uint32_t j0 = wb; j0 < we; j0 += 4) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load(j0 + u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (uint32_t e = 0; e < EPW; ++e) {
                    const uint32_t kv = (v[u] >> (e * BITS)) & MASK;
                    const bool lt = kv < T, tie = kv == T;
                    if (lt || (tie && tie_run < budget)) emit((j0 + u) * EPW + e, lt_run + min(tie_run, budget));
                    lt_run += lt ? 1u : 0u;
                    tie_run += tie ? 1u : 0u;
                }
            }
        }
    }
    CUDDLE_STAMP(1, 7);
    __syncthreads();
}

// The same selection for keys in global memory (a query's row of the spill buffer): there a lane walking its own
// run of words makes every load instruction touch 64 cache lines, so each wave owns a contiguous segment instead and
// reads it 64 consecutive words at a time. Pass 1 counts; pass 2 ranks the keys of a step by two wave scans, and only
// in steps that hold a selected key at all (the selection is a small part of the candidates). Keys are compared field
// by field (any T). wtmp needs 2 * (blockDim.x / 64) u32. All threads call; contains two barriers.
template <class DistT, class WordPtr, class EmitFn>
__device__ __forceinline__ void wave_threshold_select(WordPtr words, uint32_t C, uint32_t T, uint32_t budget,
                                                      EmitFn emit, uint32_t *wtmp) {
    constexpr uint32_t EPW = 4 / sizeof(DistT), BITS = 8 * sizeof(DistT), MASK = (1u << BITS) - 1;
    const uint32_t nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nwords = (C + EPW - 1) / EPW;
    const uint32_t seg = (((nwords + nw - 1) / nw + 63) / 64) * 64;
    const uint32_t wb = min(w * seg, nwords), we = min(wb + seg, nwords);
    auto classify = [&](uint32_t j, uint32_t x, uint32_t &mlt, uint32_t &mtie) {
        mlt = mtie = 0;
#pragma unroll
        for (uint32_t e = 0; e < EPW; ++e) {
            const uint32_t kv = (x >> (e * BITS)) & MASK;
            const bool in = j < we && j * EPW + e < C;
            mlt |= (in && kv < T ? 1u : 0u) << e;
            mtie |= (in && kv == T ? 1u : 0u) << e;
        }
    };
    constexpr int kSteps = 4; // loads of four steps are issued before the first is used
    uint32_t n_lt = 0, n_tie = 0;
    for (uint32_t j0 = wb; j0 < we; j0 += 64 * kSteps) {
        uint32_t x[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) x[u] = words[min(j0 + 64 * u + lane, nwords - 1)];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            uint32_t mlt, mtie;
            classify(j0 + 64 * u + lane, x[u], mlt, mtie);
            n_lt += __popc(mlt);
            n_tie += __popc(mtie);
        }
    }
    const uint32_t tot_lt = wave_inclusive_sum(n_lt), tot_tie = wave_inclusive_sum(n_tie);
    if (lane == 63) {
        wtmp[w] = tot_lt;
        wtmp[nw + w] = tot_tie;
    }
    __syncthreads();
    uint32_t lt_run = 0, tie_run = 0; // selected / tied keys before this wave's current step
    for (uint32_t v = 0; v < w; ++v) {
        lt_run += wtmp[v];
        tie_run += wtmp[nw + v];
    }
    for (uint32_t j0 = wb; j0 < we; j0 += 64 * kSteps) {
        uint32_t x[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) x[u] = words[min(j0 + 64 * u + lane, nwords - 1)];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const uint32_t j = j0 + 64 * u + lane;
            uint32_t mlt, mtie;
            classify(j, x[u], mlt, mtie);
            if (__ballot((mlt | mtie) != 0) == 0) continue; // (wave-uniform)
            const uint32_t c_lt = __popc(mlt), c_tie = __popc(mtie);
            const uint32_t i_lt = wave_inclusive_sum(c_lt), i_tie = wave_inclusive_sum(c_tie);
            uint32_t my_lt = lt_run + i_lt - c_lt, my_tie = tie_run + i_tie - c_tie;
#pragma unroll
            for (uint32_t e = 0; e < EPW; ++e) {
                if ((mlt >> e) & 1u) {
                    emit(j * EPW + e, my_lt + min(my_tie, budget));
                    ++my_lt;
                } else if ((mtie >> e) & 1u) {
                    if (my_tie < budget) emit(j * EPW + e, my_lt + my_tie);
                    ++my_tie;
                }
            }
            lt_run += (uint32_t)__builtin_amdgcn_readlane((int)i_lt, 63);
            tie_run += (uint32_t)__builtin_amdgcn_readlane((int)i_tie, 63);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// exact distances in the reference's accumulation order
// ---------------------------------------------------------------------------------------------

// Value of the lane whose number differs in bit 0 / bit 1 (the other lanes of a 4-lane group),
// as a DPP quad permutation: no LDS round trip, and the compiler can fold it into the add.
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
}

// (sum0+sum1)+(sum2+sum3) over the four lanes' accumulators, lane-wise, then the pairwise
// horizontal add (u0+u1)+(u2+u3): every lane of the group returns the same value.
__device__ __forceinline__ float group4_reduce(float4 a) {
    a.x += quad_xor1(a.x); a.y += quad_xor1(a.y); a.z += quad_xor1(a.z); a.w += quad_xor1(a.w);
    a.x += quad_xor2(a.x); a.y += quad_xor2(a.y); a.z += quad_xor2(a.z); a.w += quad_xor2(a.w);
    return (a.x + a.y) + (a.z + a.w);
}

__device__ __forceinline__ float4 ld4(const float *p, bool vec) {
    if (vec) return *reinterpret_cast<const float4 *>(p);
    return make_float4(p[0], p[1], p[2], p[3]);
}

// Tails shared by the dot / L2 routines after the 16-wide main loop
// (simd_neon.rs:54-67 and :122-137): 4-element chunks with a separately rounded multiply and a
// pairwise add, then single elements.
template <bool L2>
__device__ __forceinline__ float neon_tail(float r, const float *a, const float *b, uint32_t i, uint32_t n) {
    for (; i + 4 <= n; i += 4) {
        float x0, x1, x2, x3;
        if (L2) {
            float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
            x0 = d0 * d0; x1 = d1 * d1; x2 = d2 * d2; x3 = d3 * d3;
        } else {
            x0 = a[i] * b[i]; x1 = a[i + 1] * b[i + 1]; x2 = a[i + 2] * b[i + 2]; x3 = a[i + 3] * b[i + 3];
        }
        r += (x0 + x1) + (x2 + x3);
    }
    for (; i < n; ++i) {
        if (L2) {
            float d = a[i] - b[i];
            r += d * d;
        } else {
            r += a[i] * b[i];
        }
    }
    return r;
}

// dot_product_neon / l2_squared_neon (simd_neon.rs:13-70 / :77-140) computed by a group of four
// consecutive lanes: lane `sub` (0..3) owns the reference's NEON accumulator sum<sub> (four f32
// lanes, elements 16t+4sub..+3), runs the same fused-multiply-add chain over t, and the four
// accumulators are combined (sum0+sum1)+(sum2+sum3) lane-wise, then (u0+u1)+(u2+u3) — the
// vaddvq_f32 pairwise order. All four lanes return the result. a may point to LDS.
template <bool L2>
__device__ __forceinline__ float neon16_group4(const float *a, const float *b, uint32_t n, uint32_t sub, bool vec) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t nblk = n >> 4;
    const float *pa = a + 4 * sub, *pb = b + 4 * sub;
#pragma unroll 4
    for (uint32_t t = 0; t < nblk; ++t) {
        float4 x = ld4(pa + 16 * t, vec), y = ld4(pb + 16 * t, vec);
        if (L2) {
            float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc.x = __builtin_fmaf(d0, d0, acc.x);
            acc.y = __builtin_fmaf(d1, d1, acc.y);
            acc.z = __builtin_fmaf(d2, d2, acc.z);
            acc.w = __builtin_fmaf(d3, d3, acc.w);
        } else {
            acc.x = __builtin_fmaf(x.x, y.x, acc.x);
            acc.y = __builtin_fmaf(x.y, y.y, acc.y);
            acc.z = __builtin_fmaf(x.z, y.z, acc.z);
            acc.w = __builtin_fmaf(x.w, y.w, acc.w);
        }
    }
    return neon_tail<L2>(group4_reduce(acc), a, b, nblk << 4, n);
}

// neon16_group4 / neon16_group4_dot2 for a row b that comes from far away (a random row of the fp32 vectors in HBM):
// U of the row's 16-byte pieces per lane are requested before the first is used, so a 3840-byte row takes 4 trips to
// memory instead of 15 and its pieces reach the DRAM page together. Same chains, same values.
template <bool L2, bool BB, int U>
__device__ __forceinline__ void neon16_group4_far(const float *a, const float *b, uint32_t n, uint32_t sub, bool vec, float &ab,
                                                  float &bb) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acb = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t nblk = n >> 4;
    const float *pa = a + 4 * sub, *pb = b + 4 * sub;
    auto step = [&](const float4 x, const float4 y) {
        if (L2) {
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc.x = __builtin_fmaf(d0, d0, acc.x);
            acc.y = __builtin_fmaf(d1, d1, acc.y);
            acc.z = __builtin_fmaf(d2, d2, acc.z);
            acc.w = __builtin_fmaf(d3, d3, acc.w);
        } else {
            acc.x = __builtin_fmaf(x.x, y.x, acc.x);
            acc.y = __builtin_fmaf(x.y, y.y, acc.y);
            acc.z = __builtin_fmaf(x.z, y.z, acc.z);
            acc.w = __builtin_fmaf(x.w, y.w, acc.w);
        }
        if (BB) {
            acb.x = __builtin_fmaf(y.x, y.x, acb.x);
            acb.y = __builtin_fmaf(y.y, y.y, acb.y);
            acb.z = __builtin_fmaf(y.z, y.z, acb.z);
            acb.w = __builtin_fmaf(y.w, y.w, acb.w);
        }
    };
    uint32_t t = 0;
    for (; t + U <= nblk; t += U) {
        float4 y[U];
#pragma unroll
        for (int k = 0; k < U; ++k) y[k] = ld4(pb + 16 * (t + k), vec);
#pragma unroll
        for (int k = 0; k < U; ++k) step(ld4(pa + 16 * (t + k), vec), y[k]);
    }
#pragma unroll 4
    for (; t < nblk; ++t) step(ld4(pa + 16 * t, vec), ld4(pb + 16 * t, vec));
    ab = neon_tail<L2>(group4_reduce(acc), a, b, nblk << 4, n);
    if (BB) bb = neon_tail<false>(group4_reduce(acb), b, b, nblk << 4, n);
}

// The same routines computed by one lane (16 running partial sums).
template <bool L2>
__device__ __forceinline__ float neon16_serial(const float *a, const float *b, uint32_t n) {
    float s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0.f;
    const uint32_t nblk = n >> 4;
    for (uint32_t t = 0; t < nblk; ++t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float x = a[16 * t + j], y = b[16 * t + j];
            if (L2) {
                float d = x - y;
                s[j] = __builtin_fmaf(d, d, s[j]);
            } else {
                s[j] = __builtin_fmaf(x, y, s[j]);
            }
        }
    }
    float u[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) u[l] = (s[l] + s[4 + l]) + (s[8 + l] + s[12 + l]);
    float r = (u[0] + u[1]) + (u[2] + u[3]);
    return neon_tail<L2>(r, a, b, nblk << 4, n);
}

// One target of l2_squared_batch4_neon (simd_neon.rs:160-227): a single 4-lane accumulator over
// elements i = l (mod 4), vaddvq, then a scalar tail.
__device__ __forceinline__ float l2_batch4_serial(const float *q, const float *t, uint32_t n) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {
        float d0 = q[i] - t[i], d1 = q[i + 1] - t[i + 1], d2 = q[i + 2] - t[i + 2], d3 = q[i + 3] - t[i + 3];
        a0 = __builtin_fmaf(d0, d0, a0);
        a1 = __builtin_fmaf(d1, d1, a1);
        a2 = __builtin_fmaf(d2, d2, a2);
        a3 = __builtin_fmaf(d3, d3, a3);
    }
    float r = (a0 + a1) + (a2 + a3);
    for (; i < n; ++i) {
        float d = q[i] - t[i];
        r += d * d;
    }
    return r;
}

// l2_batch4_serial by a 4-lane group: lane `sub` keeps lane `sub` of the accumulator (elements i = sub mod 4, in
// order), the quad adds them as vaddvq does, (a0 + a1) + (a2 + a3), and the tail is added in order. Every lane returns
// the result. All four lanes of the group call.
__device__ __forceinline__ float l2_batch4_group4(const float *q, const float *t, uint32_t n, uint32_t sub) {
    float a = 0.f;
    const uint32_t n4 = n & ~3u;
    uint32_t i = sub;
    for (; i + 28 < n4; i += 32) { // eight steps of the chain with their sixteen loads in flight together
        float qv[8], tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            qv[k] = q[i + 4 * k];
            tv[k] = t[i + 4 * k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = qv[k] - tv[k];
            a = __builtin_fmaf(d, d, a);
        }
    }
    for (; i < n4; i += 4) {
        const float d = q[i] - t[i];
        a = __builtin_fmaf(d, d, a);
    }
    a += quad_xor1(a); // lanes 0, 1: a0 + a1; lanes 2, 3: a2 + a3
    float r = a + quad_xor2(a); // (a0 + a1) + (a2 + a3) in every lane (the additions commute)
    for (uint32_t i = n4; i < n; ++i) {
        const float d = q[i] - t[i];
        r += d * d;
    }
    return r;
}

// distance() (src/distance/mod.rs:23-34) by a 4-lane group. a = query side, b = stored side.
__device__ __forceinline__ float distance_group4(const float *a, const float *b, uint32_t n, uint32_t metric,
                                                 uint32_t sub, bool vec) {
    if (metric == kMetricL2) return neon16_group4<true>(a, b, n, sub, vec);
    float dot = neon16_group4<false>(a, b, n, sub, vec);
    if (metric == kMetricDot) return -dot;
    float na = sqrtf(neon16_group4<false>(a, a, n, sub, vec));
    float nb = sqrtf(neon16_group4<false>(b, b, n, sub, vec));
    return 1.0f - (dot / (na * nb));
}

// neon16_group4 for NQ query vectors (LDS, qstride floats apart) against one stored vector b: the
// stored vector is read once, each query keeps its own accumulators; every result is the one
// neon16_group4 gives for that pair.
template <bool L2, int NQ>
__device__ __forceinline__ void neon16_group4_multi(const float *qs, uint32_t qstride, const float *b, uint32_t n,
                                                    uint32_t sub, bool vec, float *out) {
    float4 acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t nblk = n >> 4;
    const float *pa = qs + 4 * sub, *pb = b + 4 * sub;
#pragma unroll 2
    for (uint32_t t = 0; t < nblk; ++t) {
        const float4 y = ld4(pb + 16 * t, vec);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float4 x = ld4(pa + (uint32_t)j * qstride + 16 * t, vec);
            if (L2) {
                const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                acc[j].x = __builtin_fmaf(d0, d0, acc[j].x);
                acc[j].y = __builtin_fmaf(d1, d1, acc[j].y);
                acc[j].z = __builtin_fmaf(d2, d2, acc[j].z);
                acc[j].w = __builtin_fmaf(d3, d3, acc[j].w);
            } else {
                acc[j].x = __builtin_fmaf(x.x, y.x, acc[j].x);
                acc[j].y = __builtin_fmaf(x.y, y.y, acc[j].y);
                acc[j].z = __builtin_fmaf(x.z, y.z, acc[j].z);
                acc[j].w = __builtin_fmaf(x.w, y.w, acc[j].w);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        out[j] = neon_tail<L2>(group4_reduce(acc[j]), qs + (uint32_t)j * qstride, b, nblk << 4, n);
    }
}

// distance() of NQ queries (LDS) to one stored vector b by a 4-lane group; out[j] equals
// distance_group4(query j, b).
template <int NQ>
__device__ __forceinline__ void distance_group4_multi(const float *qs, uint32_t qstride, const float *b, uint32_t n,
                                                      uint32_t metric, uint32_t sub, bool vec, float *out) {
    if (metric == kMetricL2) {
        neon16_group4_multi<true, NQ>(qs, qstride, b, n, sub, vec, out);
        return;
    }
    neon16_group4_multi<false, NQ>(qs, qstride, b, n, sub, vec, out);
    if (metric == kMetricDot) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) out[j] = -out[j];
        return;
    }
    const float nb = sqrtf(neon16_group4<false>(b, b, n, sub, vec));
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float *a = qs + (uint32_t)j * qstride;
        const float na = sqrtf(neon16_group4<false>(a, a, n, sub, vec));
        out[j] = 1.0f - (out[j] / (na * nb));
    }
}

// dot(a, b) and dot(b, b) of neon16_group4<false> in ONE pass over b: each sum keeps its own
// accumulators and its own order, so both equal the separately computed values.
__device__ __forceinline__ void neon16_group4_dot2(const float *a, const float *b, uint32_t n, uint32_t sub, bool vec,
                                                   float &ab, float &bb) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acb = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t nblk = n >> 4;
    const float *pa = a + 4 * sub, *pb = b + 4 * sub;
#pragma unroll 4
    for (uint32_t t = 0; t < nblk; ++t) {
        const float4 x = ld4(pa + 16 * t, vec), y = ld4(pb + 16 * t, vec);
        acc.x = __builtin_fmaf(x.x, y.x, acc.x);
        acc.y = __builtin_fmaf(x.y, y.y, acc.y);
        acc.z = __builtin_fmaf(x.z, y.z, acc.z);
        acc.w = __builtin_fmaf(x.w, y.w, acc.w);
        acb.x = __builtin_fmaf(y.x, y.x, acb.x);
        acb.y = __builtin_fmaf(y.y, y.y, acb.y);
        acb.z = __builtin_fmaf(y.z, y.z, acb.z);
        acb.w = __builtin_fmaf(y.w, y.w, acb.w);
    }
    ab = neon_tail<false>(group4_reduce(acc), a, b, nblk << 4, n);
    bb = neon_tail<false>(group4_reduce(acb), b, b, nblk << 4, n);
}

// distance() by a 4-lane group when the query-side norm na = sqrtf(dot(a, a)) is already known
// (it is the same for every stored vector the query meets): cosine reads b once instead of twice
// and skips the query's own sum. Same value as distance_group4.
__device__ __forceinline__ float distance_group4_qn(const float *a, const float *b, uint32_t n, uint32_t metric,
                                                    uint32_t sub, bool vec, float na) {
    if (metric == kMetricL2) return neon16_group4<true>(a, b, n, sub, vec);
    if (metric == kMetricDot) return -neon16_group4<false>(a, b, n, sub, vec);
    float ab, bb;
    neon16_group4_dot2(a, b, n, sub, vec, ab, bb);
    return 1.0f - (ab / (na * sqrtf(bb)));
}

// qnorm[q] = sqrtf(dot(q, q)) in the arithmetic of distance(): one 4-lane group per query.
__global__ void query_norm_kernel(const float *queries, uint32_t nq, uint32_t dim, float *qnorm) {
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, sub = threadIdx.x & 3;
    const bool act = g < nq;
    const float *q = queries + (uint64_t)(act ? g : 0) * dim;
    const float s = neon16_group4<false>(q, q, dim, sub, (dim & 3) == 0);
    if (act && sub == 0) qnorm[g] = sqrtf(s);
}

// distance_group4_qn for a far row (see neon16_group4_far).
template <int U>
__device__ __forceinline__ float distance_group4_qn_far(const float *a, const float *b, uint32_t n, uint32_t metric, uint32_t sub,
                                                        bool vec, float na) {
    float ab, bb = 0.f;
    if (metric == kMetricL2) {
        neon16_group4_far<true, false, U>(a, b, n, sub, vec, ab, bb);
        return ab;
    }
    if (metric == kMetricDot) {
        neon16_group4_far<false, false, U>(a, b, n, sub, vec, ab, bb);
        return -ab;
    }
    neon16_group4_far<false, true, U>(a, b, n, sub, vec, ab, bb);
    return 1.0f - (ab / (na * sqrtf(bb)));
}

__device__ __forceinline__ float distance_serial(const float *a, const float *b, uint32_t n, uint32_t metric) {
    if (metric == kMetricL2) return neon16_serial<true>(a, b, n);
    float dot = neon16_serial<false>(a, b, n);
    if (metric == kMetricDot) return -dot;
    float na = sqrtf(neon16_serial<false>(a, a, n));
    float nb = sqrtf(neon16_serial<false>(b, b, n));
    return 1.0f - (dot / (na * nb));
}

// ---------------------------------------------------------------------------------------------
// stand-alone pair kernels (stage entry points)
// ---------------------------------------------------------------------------------------------

// distance() for n independent pairs; one 4-lane group per pair.
__global__ void distance_pairs_kernel(const float *a, const float *b, uint64_t n, uint32_t dim, uint32_t metric,
                                      float *out) {
    uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    uint32_t sub = threadIdx.x & 3;
    bool act = g < n;
    uint64_t gg = act ? g : 0; // inactive groups still run the shuffles on pair 0
    bool vec = (dim & 3) == 0;
    float d = distance_group4(a + gg * dim, b + gg * dim, dim, metric, sub, vec);
    if (act && sub == 0) out[g] = d;
}

// The work of block `bid` of binary_quantize_kernel (256 threads: one wave per (vector, word)).
__device__ __forceinline__ void binary_quantize_block(uint32_t bid, const float *v, uint64_t n, uint32_t dim, uint32_t words,
                                                      float thr, uint64_t *codes) {
    const uint64_t t = (uint64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n * words) return;
    const uint32_t lane = threadIdx.x & 63, w = (uint32_t)(t % words), i = w * 64 + lane;
    const bool bit = i < dim && v[(t / words) * dim + i] > thr;
    const uint64_t bits = __ballot(bit);
    if (lane == 0) codes[t] = bits;
}

// BinaryQuantizer::quantize (binary.rs:144-159): one wave per (vector, word); lane i tests value
// 64 w + i (strict >), the wave's ballot is the word (bit i = lane i, pad bits 0).
// grid = ceil(n * words / 4), block 256.
__global__ void __launch_bounds__(256)
binary_quantize_kernel(const float *v, uint64_t n, uint32_t dim, uint32_t words, float thr, uint64_t *codes) {
    binary_quantize_block(blockIdx.x, v, n, dim, words, thr, codes);
}

// hamming_distance (binary.rs:42-54) for n pairs.
__global__ void hamming_pairs_kernel(const uint64_t *a, const uint64_t *b, uint64_t n, uint32_t words, uint32_t *out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t d = 0;
    for (uint32_t w = 0; w < words; ++w) d += __popcll(a[t * words + w] ^ b[t * words + w]);
    out[t] = d;
}

// ---------------------------------------------------------------------------------------------
// index set-up kernels
// ---------------------------------------------------------------------------------------------

// pos_leaf[p] = node whose vector range contains leaf-contiguous position p (binary search in vec_off).
__global__ void pos_leaf_kernel(const uint32_t *vec_off, uint32_t n_nodes, uint64_t n_vectors, uint32_t *pos_leaf) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_vectors) return;
    uint32_t lo = 0, hi = n_nodes; // vec_off[lo] <= p < vec_off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (vec_off[mid] <= p) lo = mid; else hi = mid;
    }
    pos_leaf[p] = lo;
}

// acc[i] += sum over this block's slice of the listed nodes (nodes == null: rows 0..n) of
// centroid[node][i] (double); grid = (ceil(dim/64), slices), block = 64.
__global__ void level_sum_kernel(const float *centroids, const uint32_t *nodes, uint32_t n, uint32_t dim, double *acc) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= dim) return;
    double s = 0.0;
    for (uint32_t j = blockIdx.y; j < n; j += gridDim.y) s += (double)centroids[(uint64_t)(nodes ? nodes[j] : j) * dim + i];
    atomicAdd(&acc[i], s);
}
__global__ void level_mean_kernel(const double *acc, uint32_t n, uint32_t dim, float *mu) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dim) mu[i] = (float)(acc[i] / (double)n);
}

// Re-lay vector-major binary codes [n_vectors][words] into the per-leaf planar layout the scan
// streams: for a leaf with n vectors starting at 16-byte unit code_off, plane p (p < words/2)
// holds words (2p, 2p+1) of vector i at 16-byte slot p*n + i; when `words` is odd a last plane of
// 8-byte entries holds word words-1 of vector i. A wave reading one plane for 64 consecutive
// vectors issues one fully coalesced 1 KiB load. One thread per (leaf-contiguous position).
__global__ void relayout_codes_kernel(const uint64_t *src, uint64_t *dst, const uint32_t *vec_off,
                                      const uint32_t *code_off, const uint32_t *pos_leaf, uint64_t n_vectors,
                                      uint32_t words) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_vectors) return;
    uint32_t leaf = pos_leaf[p];
    uint32_t i = (uint32_t)(p - vec_off[leaf]);
    uint32_t n = vec_off[leaf + 1] - vec_off[leaf];
    uint64_t *base = dst + (uint64_t)code_off[leaf] * 2;
    const uint64_t *s = src + p * words;
    uint32_t pairs = words >> 1;
    for (uint32_t q = 0; q < pairs; ++q) {
        base[((uint64_t)q * n + i) * 2] = s[2 * q];
        base[((uint64_t)q * n + i) * 2 + 1] = s[2 * q + 1];
    }
    if (words & 1) base[(uint64_t)pairs * n * 2 + i] = s[words - 1];
}

// RaBitQ codes: vector-major [n_vectors][words32] u32 -> per leaf word-major [words32][n].
__global__ void relayout_rabitq_kernel(const uint32_t *src, uint32_t *dst, const uint32_t *vec_off,
                                       const uint32_t *pos_leaf, uint64_t n_vectors, uint32_t words32) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_vectors) return;
    uint32_t leaf = pos_leaf[p];
    uint64_t beg = vec_off[leaf];
    uint32_t i = (uint32_t)(p - beg);
    uint32_t n = vec_off[leaf + 1] - vec_off[leaf];
    uint32_t *base = dst + beg * words32;
    for (uint32_t w = 0; w < words32; ++w) base[(uint64_t)w * n + i] = src[p * words32 + w];
}

// Vector-major copy of the RaBitQ codes with rows padded to whole 16-byte pieces.
__global__ void pad_rabitq_rows_kernel(const uint32_t *src, uint32_t *dst, uint64_t n_vectors, uint32_t words32, uint32_t vm_words,
                                       uint64_t t0) {
    const uint64_t t = t0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_vectors * vm_words) return;
    const uint64_t p = t / vm_words;
    const uint32_t w = (uint32_t)(t % vm_words);
    dst[t] = w < words32 ? src[p * words32 + w] : 0u;
}

// ---------------------------------------------------------------------------------------------
// Hamming leaf scan + phase-1 selection (hierarchical.rs:591-655)
// ---------------------------------------------------------------------------------------------

// Hamming distance of vector i of a leaf (n vectors, planar block at `planes`) to the query code
// qw[0..words) (LDS). 128-bit loads; consecutive lanes read consecutive 16-byte slots.
__device__ __forceinline__ uint32_t leaf_hamming(const uint4 *planes, uint32_t n, uint32_t i, const uint64_t *qw,
                                                 uint32_t words) {
    const uint32_t pairs = words >> 1;
    uint32_t h = 0;
    for (uint32_t p = 0; p < pairs; ++p) {
        uint4 c = planes[(uint64_t)p * n + i];
        uint64_t a = ((uint64_t)c.y << 32) | c.x, b = ((uint64_t)c.w << 32) | c.z;
        h += __popcll(a ^ qw[2 * p]) + __popcll(b ^ qw[2 * p + 1]);
    }
    if (words & 1) {
        const uint2 *tail = reinterpret_cast<const uint2 *>(planes + (uint64_t)pairs * n);
        uint2 c = tail[i];
        uint64_t a = ((uint64_t)c.y << 32) | c.x;
        h += __popcll(a ^ qw[words - 1]);
    }
    return h;
}

// LDS carve-up shared by the scan kernels.
struct ScanLds {
    uint32_t *hist;     // [dim + 2]
    uint32_t *lbeg;     // [lcap]      first leaf-contiguous position of each leaf slot
    uint32_t *lcode;    // [lcap + 1]  16-byte unit offset of the leaf's planar codes (one spare: reused for prefix sums)
    uint32_t *lscan;    // [lcap + 1]  exclusive prefix of leaf sizes = scan position of the leaf
    uint32_t *ltile;    // [lcap + 1]  exclusive prefix of the leaves' 64-vector tile counts
    uint32_t *tiles;    // [tcap]      (leaf slot << 12) | tile number inside the leaf
    uint32_t *tinfo;    // [tcap]      (scan position of the tile's first vector << 7) | vectors in the tile
    uint32_t *tcode;    // [tcap]      16-byte unit offset of the tile's first vector in plane 0 (optional)
    uint32_t *wtmp;     // [32]
    uint32_t *misc;     // [8]
    uint64_t *qw;       // [words]
    uint16_t *dist;     // [dist_cap]
};

__device__ __host__ __forceinline__ uint32_t scan_pad4(uint32_t v) { return (v + 3) & ~3u; }

__device__ __forceinline__ ScanLds scan_lds_carve(unsigned char *smem, uint32_t dim, uint32_t words, uint32_t lcap,
                                                  uint32_t tcap, bool with_tcode = false) {
    ScanLds s;
    uint32_t *p = reinterpret_cast<uint32_t *>(smem);
    s.hist = p; p += scan_pad4(dim + 2);
    s.lbeg = p; p += scan_pad4(lcap);
    s.lcode = p; p += scan_pad4(lcap + 1);
    s.lscan = p; p += scan_pad4(lcap + 1);
    s.ltile = p; p += scan_pad4(lcap + 1);
    s.tiles = p; p += scan_pad4(tcap);
    s.tinfo = p; p += scan_pad4(tcap);
    s.tcode = with_tcode ? p : nullptr; p += with_tcode ? scan_pad4(tcap) : 0;
    s.wtmp = p; p += 32;
    s.misc = p; p += 8;
    s.qw = reinterpret_cast<uint64_t *>(p); p += 2 * ((words + 1) & ~1u);
    s.dist = reinterpret_cast<uint16_t *>(p);
    return s;
}

inline size_t scan_lds_bytes(uint32_t dim, uint32_t words, uint32_t lcap, uint32_t tcap, uint32_t dist_cap,
                             bool with_tcode = false) {
    size_t u32s = scan_pad4(dim + 2) + (size_t)scan_pad4(lcap) + 3 * (size_t)scan_pad4(lcap + 1) +
                  (with_tcode ? 3 : 2) * (size_t)scan_pad4(tcap) + 32 +
                  8 + 2 * (size_t)((words + 1) & ~1u);
    return u32s * 4 + (size_t)dist_cap * 2;
}

// Loads the query's leaf descriptors into LDS, prefix-sums the leaf sizes (scan positions) and
// cuts every leaf into tiles of 64 vectors (the unit one wave processes per step). Returns the
// number of candidates C; *ntiles receives the tile count (capped at tcap). All threads call.
__device__ __forceinline__ uint32_t scan_load_leaves(const IndexDev &ix, const ScanLds &s, const uint32_t *lv,
                                                     uint32_t nl, uint32_t tcap, uint32_t *ntiles) {
    for (uint32_t t = threadIdx.x; t < nl; t += blockDim.x) {
        uint32_t node = lv[t];
        uint32_t b = ix.vec_off[node], e = ix.vec_off[node + 1];
        s.lbeg[t] = b;
        s.lscan[t] = e - b;
        s.ltile[t] = (e - b + 63) >> 6;
        s.lcode[t] = ix.code_off[node];
    }
    __syncthreads();
    wave0_exclusive_scan_lds(s.lscan, nl);
    __syncthreads();
    wave0_exclusive_scan_lds(s.ltile, nl);
    __syncthreads();
    if (tcap) {
        for (uint32_t t = threadIdx.x; t < nl; t += blockDim.x) {
            const uint32_t tb = s.ltile[t], te = s.ltile[t + 1];
            const uint32_t base = s.lscan[t], n = s.lscan[t + 1] - base;
            for (uint32_t c = tb; c < te && c < tcap; ++c) {
                const uint32_t first = (c - tb) << 6;
                s.tiles[c] = (t << 12) | (c - tb);
                s.tinfo[c] = ((base + first) << 7) | min(64u, n - first);
                if (s.tcode) s.tcode[c] = s.lcode[t] + first;
            }
        }
        __syncthreads();
    }
    if (ntiles) *ntiles = min(s.ltile[nl], tcap);
    return s.lscan[nl];
}

// The same from the tables scan_tables_kernel prepared for this block (tb: four rows of tstride u32), in
// two steps. scan_copy_tables: one round of coalesced copies whose length does not depend on anything
// loaded before (rows are copied whole, lcap + 1 entries; the query number and its leaf count come with
// them), then a barrier. scan_build_tiles: cuts the leaves into 64-vector tiles, then a barrier; returns
// the number of candidates. All threads call both.
__device__ __forceinline__ void scan_copy_tables(const ScanLds &s, const uint32_t *tb, uint32_t tstride, uint32_t lcap,
                                                 uint32_t *q_out, uint32_t *nl_out) {
    for (uint32_t t = threadIdx.x; t <= lcap; t += blockDim.x) {
        s.lscan[t] = tb[tstride + t];
        s.ltile[t] = tb[2 * tstride + t];
        if (t < lcap) {
            s.lbeg[t] = tb[t];
            s.lcode[t] = tb[3 * tstride + t];
        }
    }
    if (threadIdx.x < 2) s.misc[4 + threadIdx.x] = tb[tstride - 2 + threadIdx.x];
    __syncthreads();
    *q_out = s.misc[4];
    *nl_out = min(s.misc[5], lcap);
}

__device__ __forceinline__ uint32_t scan_build_tiles(const ScanLds &s, uint32_t nl, uint32_t tcap, uint32_t *ntiles) {
    for (uint32_t t = threadIdx.x; t < nl; t += blockDim.x) {
        const uint32_t tb0 = s.ltile[t], te = s.ltile[t + 1];
        const uint32_t base = s.lscan[t], n = s.lscan[t + 1] - base;
        for (uint32_t c = tb0; c < te && c < tcap; ++c) {
            const uint32_t first = (c - tb0) << 6;
            s.tiles[c] = (t << 12) | (c - tb0);
            s.tinfo[c] = ((base + first) << 7) | min(64u, n - first);
            if (s.tcode) s.tcode[c] = s.lcode[t] + first;
        }
    }
    __syncthreads();
    *ntiles = min(s.ltile[nl], tcap);
    return s.lscan[nl];
}

// Leaf slot that contains scan position i (largest s with lscan[s] <= i, skipping empty leaves).
__device__ __forceinline__ uint32_t scan_slot_of(const uint32_t *lscan, uint32_t nl, uint32_t i) {
    uint32_t lo = 0, hi = nl; // invariant: lscan[lo] <= i < lscan[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (lscan[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// Phase 1 of the scan for codes of 2*PAIRS (+1 if TAIL) words: each wave takes U tiles at a time
// and issues all of their 128-bit loads (U * PAIRS per lane, every one a fully coalesced
// wave-wide read of consecutive 16-byte slots) before the first XOR/popcount, so that one wave
// keeps U * PAIRS KiB in flight. Distances are stored by scan position.
// DistPtr: DistT * in the address space the distances live in (LDS or the global spill buffer).
template <int PAIRS, bool TAIL, int U, class DistT, class DistPtr>
__device__ __forceinline__ void scan_tiles(const IndexDev &ix, const ScanLds &s, uint32_t ntiles, DistPtr dist) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (PAIRS == 1 && !TAIL) {
        // two-word codes: one 16-byte slot per vector, addressed through the per-tile tables alone
        const uint64_t q0 = s.qw[0], q1 = s.qw[1];
        for (uint32_t t0 = w * U; t0 < ntiles; t0 += nw * U) {
            uint4 c[U];
            uint32_t ti[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t t = t0 + u;
                ti[u] = t < ntiles ? s.tinfo[t] : 0u;
                const uint32_t co = s.tcode[t < ntiles ? t : t0];
                c[u] = lane < (ti[u] & 127u) ? ix.codes[(uint64_t)co + lane] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t a = ((uint64_t)c[u].y << 32) | c[u].x, b = ((uint64_t)c[u].w << 32) | c[u].z;
                const uint32_t h = __popcll(a ^ q0) + __popcll(b ^ q1);
                if (lane < (ti[u] & 127u)) {
                    dist[(ti[u] >> 7) + lane] = (DistT)h;
                    atomicAdd(&s.hist[h], 1u);
                }
            }
        }
        return;
    }
    for (uint32_t t0 = w * U; t0 < ntiles; t0 += nw * U) {
        uint4 c[U][PAIRS > 0 ? PAIRS : 1];
        uint2 ct[U];
        uint32_t pos[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u;
            const uint32_t tile = s.tiles[t < ntiles ? t : t0];
            const uint32_t sl = tile >> 12, i = ((tile & 4095u) << 6) + lane;
            const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
            act[u] = t < ntiles && i < n;
            pos[u] = base + i;
            const uint4 *planes = ix.codes + s.lcode[sl];
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) c[u][p] = act[u] ? planes[(uint64_t)p * n + i] : make_uint4(0, 0, 0, 0);
            if (TAIL) {
                const uint2 *tail = reinterpret_cast<const uint2 *>(planes + (uint64_t)PAIRS * n);
                ct[u] = act[u] ? tail[i] : make_uint2(0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = 0;
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                const uint64_t a = ((uint64_t)c[u][p].y << 32) | c[u][p].x, b = ((uint64_t)c[u][p].w << 32) | c[u][p].z;
                h += __popcll(a ^ s.qw[2 * p]) + __popcll(b ^ s.qw[2 * p + 1]);
            }
            if (TAIL) h += __popcll((((uint64_t)ct[u].y << 32) | ct[u].x) ^ s.qw[2 * PAIRS]);
            if (act[u]) {
                dist[pos[u]] = (DistT)h;
                atomicAdd(&s.hist[h], 1u);
            }
        }
    }
}

// Candidates per query = sum of the sizes of its reached leaves (one wave per query).
// zero != null: the launch also clears zero[0 .. n_zero) (the per-leaf pair counters scan_tables_kernel fills next).
__global__ void __launch_bounds__(256)
cand_count_kernel(IndexDev ix, const uint32_t *leaves, uint32_t lstride, const uint32_t *lcount, uint32_t nq,
                  uint32_t *cand_count, uint32_t *zero = nullptr, uint32_t n_zero = 0, uint32_t *loc_key = nullptr) {
    if (zero)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) zero[i] = 0;
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const uint32_t nl = min(lcount[q], lstride);
    uint32_t c = 0;
    for (uint32_t t = lane; t < nl; t += 64) {
        uint32_t node = leaves[(uint64_t)q * lstride + t];
        c += ix.vec_off[node + 1] - ix.vec_off[node];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) {
        cand_count[q] = c;
        // where the query's first leaf lies in the leaf-contiguous arrays: queries with neighbouring keys read
        // neighbouring rows (see scan_tables_kernel's rr_order)
        if (loc_key) loc_key[q] = nl ? ix.vec_off[leaves[(uint64_t)q * lstride]] : 0xFFFFFFFFu;
    }
}

// What a scan block needs to know about its query, prepared where every query of the batch is in
// flight at once instead of at the head of every scan block (dependent gathers through the node
// arrays, prefix sums). One wave per query q:
//   rank = number of queries with more candidates (ties by query number): the scan kernel's block
//          `rank` takes this query, so that the longest blocks start first and the tail of the launch
//          is made of short ones;
//   tables[rank] = four rows of tstride u32 (tstride >= lstride + 3), per slot of the query's leaf
//          list: the leaf's first position | exclusive prefix sums of the leaf sizes (entry nl = the
//          candidate count) | exclusive prefix sums of the leaves' 64-vector tile counts (entry nl =
//          the total) | the leaf's code offset. Row 0 ends with the header: [tstride - 2] = q,
//          [tstride - 1] = nl.
// pair_cnt != null: also one (query, leaf) pair per non-empty slot, counted per leaf (see leaf_hamming_kernel).
// loc_key != null: rr_order[r] = the query ranked r-th by loc_key (ties by query number): the order in which the rerank
//          takes the queries, so that blocks in flight together gather rows of the same leaves (a batch of 10 000 on
//          1M x 128-d reaches every leaf a hundred times over: 1.28 -> 1.07 ms with its queries in this order).
__global__ void __launch_bounds__(256)
scan_tables_kernel(IndexDev ix, const uint32_t *leaves, uint32_t lstride, const uint32_t *lcount, const uint32_t *cand_count,
                   uint32_t nq, uint32_t *tables, uint32_t tstride, uint32_t *pair_cnt = nullptr, uint2 *pair_info = nullptr,
                   uint32_t *pair_node = nullptr, uint32_t *pair_place = nullptr, const uint32_t *loc_key = nullptr,
                   uint32_t *rr_order = nullptr, uint32_t tables_by_key = 0) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const uint32_t mine = cand_count[q], mykey = loc_key ? loc_key[q] : 0u;
    uint32_t before = 0, kbefore = 0;
    for (uint32_t j0 = lane; j0 < nq; j0 += 64 * 8) { // eight loads in flight per lane (nq >= 1: q < nq)
        uint32_t c[8], lk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c[u] = cand_count[min(j0 + 64 * u, nq - 1)];
            lk[u] = loc_key ? loc_key[min(j0 + 64 * u, nq - 1)] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t j = j0 + 64 * u;
            before += (j < nq && (c[u] > mine || (c[u] == mine && j < q))) ? 1u : 0u;
            kbefore += (j < nq && (lk[u] < mykey || (lk[u] == mykey && j < q))) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o, 64);
        kbefore += __shfl_xor(kbefore, o, 64);
    }
    if (loc_key && lane == 0) rr_order[kbefore] = q;
    if (tables_by_key) {
        // the scan blocks in key order instead, XCD by XCD: block b runs on XCD b % 8, which takes the next run of
        // the order (as many queries as there are blocks b < nq with b % 8 == x)
        uint32_t r = kbefore, x = 0;
        for (; x < 7; ++x) {
            const uint32_t cnt = (nq - x + 7) >> 3;
            if (r < cnt) break;
            r -= cnt;
        }
        before = r * 8 + x;
    }
    const uint32_t nl = min(lcount[q], lstride);
    uint32_t *tb = tables + (uint64_t)before * 4 * tstride;
    uint32_t run_c = 0, run_t = 0;
    for (uint32_t t0 = 0; t0 < nl; t0 += 64) { // wave-uniform trip count
        const uint32_t t = t0 + lane;
        uint32_t b = 0, n = 0, co = 0;
        if (t < nl) {
            const uint32_t node = leaves[(uint64_t)q * lstride + t];
            b = ix.vec_off[node];
            n = ix.vec_off[node + 1] - b;
            co = ix.code_off[node];
        }
        const uint32_t tl = (n + 63) >> 6;
        const uint32_t ic = wave_inclusive_sum(n), it = wave_inclusive_sum(tl);
        if (t < nl) {
            tb[t] = b;
            tb[tstride + t] = run_c + ic - n;
            tb[2 * tstride + t] = run_t + it - tl;
            tb[3 * tstride + t] = co;
            if (pair_cnt) {
                // the (query, leaf) pair of this slot for the leaf-by-leaf scan: where the leaf's candidates start in the
                // query's candidate list, and the pair's place among its leaf's pairs (empty leaves: no pair)
                const uint64_t at = (uint64_t)q * lstride + t;
                const uint32_t node = leaves[at];
                pair_info[at] = n ? make_uint2(q, run_c + ic - n) : make_uint2(0xFFFFFFFFu, 0u);
                pair_node[at] = node;
                pair_place[at] = n ? atomicAdd(&pair_cnt[node], 1u) : 0u;
            }
        }
        run_c += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
        run_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
    }
    if (lane == 0) {
        tb[tstride + nl] = run_c;
        tb[2 * tstride + nl] = run_t;
        tb[tstride - 2] = q;
        tb[tstride - 1] = nl;
    }
}

// One block per query (block b takes the query scan_tables_kernel ranked b-th by work, longest first). Phase 1 streams the planar codes
// of every reached leaf (tiles of 64 vectors, lane = vector), XOR + popcount against the query
// code, and keeps each candidate's Hamming distance (DistT = u8 when dim < 255, else u16; in LDS
// when the query's candidates fit in dist_cap, else in the global spill buffer) while building a
// (dim+1)-bin histogram. Phase 2 turns the histogram into the selection threshold: T = smallest h
// with #(<= h) >= rerank_k, budget = rerank_k - #(< T). Phase 3 emits, in scan order, every
// candidate with h < T and the first `budget` with h == T: the first rerank_k elements of the
// reference's stable sort_by_key (hierarchical.rs:646-648, :930-932), i.e. a legal result of
// select_nth_unstable_by (:651), listed in scan order.
// sel_pos / sel_key: [nq][sel_stride]; gdist: [nq][gdist_stride] spill.
// One instantiation per code width (PAIRS = 0: any width, generic loop) so that each gets its own
// register allocation; gdist_v points at DistT elements.
// Occupancy: with one 16-byte slot per code (PAIRS == 1, no tail) a block's time is mostly the latency of its
// prologue and selection, so the kernel is built for 8 waves per SIMD (at most 64 registers); the wider codes
// keep more loads in flight per wave instead and take the registers they need.
template <class DistT, int PAIRS, bool TAIL, int U>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((PAIRS == 1 && !TAIL) ? 8 : 1)))
scan_select_hamming_kernel(IndexDev ix, const uint64_t *qcodes, const uint32_t *tables, uint32_t tstride, uint32_t lcap,
                           uint32_t tcap, uint32_t rerank_k, uint32_t dist_cap, void *gdist_v, uint32_t gdist_stride,
                           uint32_t *sel_pos, uint32_t *sel_key, uint32_t sel_stride, uint32_t *sel_count,
                           uint32_t *cand_count, uint32_t *ghist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t D = ix.dim, W = ix.words;
    ScanLds s = scan_lds_carve(smem, D, W, lcap, tcap, PAIRS == 1 && !TAIL);
    CUDDLE_STAMP(1, 0);
    for (uint32_t i = tid; i < D + 2; i += blockDim.x) s.hist[i] = 0;
    uint32_t ntiles = 0, q, nl;
    scan_copy_tables(s, tables + (uint64_t)blockIdx.x * 4 * tstride, tstride, lcap, &q, &nl);
    // the query's code travels while the tile table is built (W <= blockDim.x for every supported dimension)
    const uint64_t my_qw = tid < W ? qcodes[(uint64_t)q * W + tid] : 0ull;
    for (uint32_t i = blockDim.x + tid; i < W; i += blockDim.x) s.qw[i] = qcodes[(uint64_t)q * W + i];
    if (tid < W) s.qw[tid] = my_qw;
    const uint32_t C = scan_build_tiles(s, nl, tcap, &ntiles);
    if (tid == 0) cand_count[q] = C;
    if (C == 0) {
        if (tid == 0) sel_count[q] = 0;
        if (ghist)
            for (uint32_t i = tid; i <= D; i += blockDim.x) ghist[(uint64_t)q * (D + 1) + i] = 0;
        return;
    }
    CUDDLE_STAMP(1, 1);
    // histogram mode (ghist != null, leaf-sharded search): distances go to the spill buffer and the
    // kernel ends after phase 1 with the query's local histogram; the selection follows once the
    // histograms of all shards are summed (cuddle_shard.hpp)
    // The rest of the kernel, once for each address space the distances can live in: a pointer
    // that may be either makes every access a flat instruction.
    typedef __attribute__((address_space(3))) DistT LdsDist;
    typedef const __attribute__((address_space(3))) uint32_t LdsWord;
    auto body = [&](auto dist, auto dist_words) {
    // phase 1
    if (PAIRS > 0) {
        scan_tiles<PAIRS, TAIL, U, DistT>(ix, s, ntiles, dist);
    } else {
        for (uint32_t t = w; t < ntiles; t += nw) {
            const uint32_t tile = s.tiles[t], sl = tile >> 12, i = ((tile & 4095u) << 6) + lane;
            const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
            if (i < n) {
                const uint32_t h = leaf_hamming(ix.codes + s.lcode[sl], n, i, s.qw, W);
                dist[base + i] = (DistT)h;
                atomicAdd(&s.hist[h], 1u);
            }
        }
    }
    __syncthreads();
    CUDDLE_STAMP(1, 2);
    if (ghist) {
        for (uint32_t i = tid; i <= D; i += blockDim.x) ghist[(uint64_t)q * (D + 1) + i] = s.hist[i];
        return;
    }
    // phase 2
    uint32_t T, budget;
    if (C <= rerank_k) {
        T = D + 1; // every candidate is < T
        budget = 0;
    } else {
        if (w == 0) {
            uint32_t carry = 0;
            bool done = false;
            for (uint32_t b0 = 0; b0 <= D && !done; b0 += 64) {
                uint32_t b = b0 + lane;
                uint32_t v = b <= D ? s.hist[b] : 0;
                uint32_t incl = wave_inclusive_sum(v) + carry;
                uint64_t hit = __ballot(incl >= rerank_k);
                if (hit) {
                    uint32_t first = __ffsll((unsigned long long)hit) - 1;
                    uint32_t incl_f = (uint32_t)__builtin_amdgcn_readlane((int)incl, first), v_f = (uint32_t)__builtin_amdgcn_readlane((int)v, first);
                    if (lane == 0) {
                        s.misc[0] = b0 + first;
                        s.misc[1] = rerank_k - (incl_f - v_f);
                    }
                    done = true;
                }
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __syncthreads();
        T = s.misc[0];
        budget = s.misc[1];
    }
    CUDDLE_STAMP(1, 3);
    // phase 3: the selected scan positions are staged (in the tile tables, which phase 1 no longer
    // needs, or in sel_pos itself when the selection is larger than those), then mapped to
    // leaf-contiguous positions by all threads.
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    const uint32_t nsel = min(min(C, rerank_k), sel_stride);
    // (typed as an LDS pointer: a pointer that may be LDS or global turns every staged store into a flat instruction)
    LdsU32 *stg = (LdsU32 *)s.tiles;
    const bool stage_lds = nsel <= ((PAIRS == 1 && !TAIL) ? 3u : 2u) * scan_pad4(tcap);
    constexpr bool in_lds = !std::is_same<decltype(dist), DistT *>::value; // the LDS instantiation of this body
    auto to_stage = [&](uint32_t i, uint32_t rank) { if (rank < nsel) stg[rank] = i; };
    auto to_row = [&](uint32_t i, uint32_t rank) { if (rank < nsel) sp[rank] = i; };
    if constexpr (in_lds) {
        // (the wave-by-wave walk below, tried on LDS: 0.377 -> 0.399 ms per 10 000 queries at 750 of 11 460 selected)
        if (stage_lds) chunked_threshold_select<DistT>(dist_words, C, T, budget, to_stage, s.wtmp);
        else chunked_threshold_select<DistT>(dist_words, C, T, budget, to_row, s.wtmp);
    } else { // keys in the spill row: read wave by wave, coalesced
        if (stage_lds) wave_threshold_select<DistT>(dist_words, C, T, budget, to_stage, s.wtmp);
        else wave_threshold_select<DistT>(dist_words, C, T, budget, to_row, s.wtmp);
    }
    if (!stage_lds) {
        __threadfence();
        __syncthreads();
    }
    for (uint32_t r = tid; r < nsel; r += blockDim.x) {
        const uint32_t i = stage_lds ? stg[r] : __hip_atomic_load(sp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t sl = scan_slot_of(s.lscan, nl, i);
        sp[r] = s.lbeg[sl] + (i - s.lscan[sl]);
        sk[r] = dist[i];
    }
    CUDDLE_STAMP(1, 4);
    if (tid == 0) sel_count[q] = nsel;
    }; // body
    if (C <= dist_cap && !ghist) {
        body((LdsDist *)s.dist, (LdsWord *)s.dist);
    } else {
        DistT *gd = reinterpret_cast<DistT *>(gdist_v) + (uint64_t)q * gdist_stride;
        body(gd, reinterpret_cast<const uint32_t *>(gd));
    }
}

// The selection half of scan_select_hamming_kernel for distances that are already in the queries' rows of the spill
// buffer (leaf_hamming_kernel writes them there, by scan position): histogram, threshold, ordered selection, exactly
// as above (select_top_candidates, hierarchical.rs:646-651 / :930-932), reading the row wave by wave. A kernel of its
// own because it needs none of the scan's registers or LDS: 64 registers and a few KB per block, so that every query
// of a chunk of up to 1024 is resident at once (inside the scan kernel two blocks fitted a CU and a 1000-query batch
// ran as two rounds: 60 -> 30 us). ghist != null: histogram mode, as there. One block per query (block b takes the
// query ranked b-th by work); dynamic LDS = select_rows_lds_bytes(dim, lcap).
constexpr uint32_t kSelectStage = 2048; // selected scan positions staged in LDS before they are mapped (more: through sel_pos)
inline size_t select_rows_lds_bytes(uint32_t dim, uint32_t lcap) {
    return ((size_t)scan_pad4(dim + 2) + scan_pad4(lcap) + scan_pad4(lcap + 1) + kSelectStage + 32 + 8) * 4;
}
template <class DistT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8)))
select_rows_kernel(IndexDev ix, const uint32_t *tables, uint32_t tstride, uint32_t lcap, uint32_t rerank_k,
                   const DistT *gdist, uint32_t gdist_stride, uint32_t *sel_pos, uint32_t *sel_key, uint32_t sel_stride,
                   uint32_t *sel_count, uint32_t *cand_count, uint32_t *ghist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem), *lbeg = hist + scan_pad4(ix.dim + 2), *lscan = lbeg + scan_pad4(lcap);
    uint32_t *stage = lscan + scan_pad4(lcap + 1), *wtmp = stage + kSelectStage, *misc = wtmp + 32;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, D = ix.dim;
    const uint32_t *tb = tables + (uint64_t)blockIdx.x * 4 * tstride;
    for (uint32_t i = tid; i < D + 2; i += blockDim.x) hist[i] = 0;
    for (uint32_t t = tid; t <= lcap; t += blockDim.x) {
        lscan[t] = tb[tstride + t];
        if (t < lcap) lbeg[t] = tb[t];
    }
    if (tid < 2) misc[4 + tid] = tb[tstride - 2 + tid];
    __syncthreads();
    const uint32_t q = misc[4], nl = min(misc[5], lcap), C = lscan[nl];
    if (tid == 0) cand_count[q] = C;
    if (C == 0) {
        if (tid == 0) sel_count[q] = 0;
        if (ghist)
            for (uint32_t i = tid; i <= D; i += blockDim.x) ghist[(uint64_t)q * (D + 1) + i] = 0;
        return;
    }
    const DistT *row = gdist + (uint64_t)q * gdist_stride;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(row);
    constexpr uint32_t EPW = 4 / sizeof(DistT), BITS = 8 * sizeof(DistT), MASK = (1u << BITS) - 1;
    const uint32_t nwords = (C + EPW - 1) / EPW;
    for (uint32_t j0 = tid; j0 < nwords; j0 += 4 * blockDim.x) { // four loads in flight per lane
        uint32_t x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = words[min(j0 + u * blockDim.x, nwords - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = j0 + u * blockDim.x;
#pragma unroll
            for (uint32_t e = 0; e < EPW; ++e)
                if (j < nwords && j * EPW + e < C) atomicAdd(&hist[(x[u] >> (e * BITS)) & MASK], 1u);
        }
    }
    __syncthreads();
    if (ghist) {
        for (uint32_t i = tid; i <= D; i += blockDim.x) ghist[(uint64_t)q * (D + 1) + i] = hist[i];
        return;
    }
    uint32_t T, budget;
    if (C <= rerank_k) {
        T = D + 1; // every candidate is < T
        budget = 0;
    } else {
        if (w == 0) {
            uint32_t carry = 0;
            bool done = false;
            for (uint32_t b0 = 0; b0 <= D && !done; b0 += 64) {
                const uint32_t b = b0 + lane, v = b <= D ? hist[b] : 0;
                const uint32_t incl = wave_inclusive_sum(v) + carry;
                const uint64_t hit = __ballot(incl >= rerank_k);
                if (hit) {
                    const uint32_t first = __ffsll((unsigned long long)hit) - 1;
                    const uint32_t incl_f = (uint32_t)__builtin_amdgcn_readlane((int)incl, first), v_f = (uint32_t)__builtin_amdgcn_readlane((int)v, first);
                    if (lane == 0) {
                        misc[0] = b0 + first;
                        misc[1] = rerank_k - (incl_f - v_f);
                    }
                    done = true;
                }
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __syncthreads();
        T = misc[0];
        budget = misc[1];
    }
    uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    const uint32_t nsel = min(min(C, rerank_k), sel_stride);
    const bool in_lds = nsel <= kSelectStage;
    if (in_lds)
        wave_threshold_select<DistT>(words, C, T, budget, [&](uint32_t i, uint32_t rank) { if (rank < nsel) stage[rank] = i; }, wtmp);
    else
        wave_threshold_select<DistT>(words, C, T, budget, [&](uint32_t i, uint32_t rank) { if (rank < nsel) sp[rank] = i; }, wtmp);
    if (!in_lds) {
        __threadfence();
        __syncthreads();
    }
    for (uint32_t r = tid; r < nsel; r += blockDim.x) {
        const uint32_t i = in_lds ? stage[r] : __hip_atomic_load(sp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t sl = scan_slot_of(lscan, nl, i);
        sp[r] = lbeg[sl] + (i - lscan[sl]);
        sk[r] = row[i];
    }
    if (tid == 0) sel_count[q] = nsel;
}

// popcount(x) + acc in one instruction (the compiler otherwise splits it into a count and a three-way add).
__device__ __forceinline__ void popc_acc(uint32_t &acc, uint32_t x) {
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

// The (query, leaf) pairs of scan_tables_kernel in leaf order (order inside a leaf is immaterial):
// sorted[poff[leaf] + place] = (query, scan position of the leaf's first vector in the query's candidate list).
__global__ void leaf_pair_fill_kernel(const uint2 *info, const uint32_t *node_of, const uint32_t *place_of, const uint32_t *lcount,
                                      uint32_t lstride, uint32_t nq, const uint32_t *poff, uint2 *sorted) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (q >= nq || t >= min(lcount[q], lstride)) return;
    const uint64_t at = (uint64_t)q * lstride + t;
    const uint2 v = info[at];
    if (v.x == 0xFFFFFFFFu) return;
    sorted[poff[node_of[at]] + place_of[at]] = v;
}

// Work of a leaf item in 256 classes, heaviest = class 0: pairs x passes of 128 vectors over the leaf.
constexpr uint32_t kWorkBins = 256;
__device__ __forceinline__ uint32_t item_work_bin(uint32_t np, uint32_t nv) {
    return (kWorkBins - 1) - min(np * ((nv + 127) >> 7), kWorkBins - 1);
}

// Launch order of leaf_hamming_kernel's work items, heaviest first (work = pairs x 128-vector passes over the leaf): a
// launch ends with its last block, and an item of 64 pairs that starts when most of the grid is done runs on alone
// for as long as the whole rest took (measured: 97 -> 68 us for 10 000 items of 1 to 64 pairs). A counting sort by
// work class: pair_scan_kernel has counted the items per class into bins[0 .. 256); every block sums the classes
// before each one for itself, and an item takes the next free place of its class (bins[256 ..], zero at launch).
// Order inside a class is immaterial. grid = ceil(max items / 256), block 256.
__global__ void __launch_bounds__(256)
item_scatter_kernel(IndexDev ix, const uint32_t *n_items, const uint4 *items, uint32_t *bins, uint32_t *order) {
    __shared__ uint32_t start[kWorkBins];
    const uint32_t tid = threadIdx.x, n = *n_items;
    if (blockIdx.x * 256 >= n) return;
    {
        const uint32_t h = bins[tid], incl = wave_inclusive_sum(h);
        __shared__ uint32_t wsum[4];
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t v = 0; v < (tid >> 6); ++v) before += wsum[v];
        start[tid] = before + incl - h;
    }
    __syncthreads();
    // places inside a class: by LDS counter within the block, one global add per class and block for the block's base
    __shared__ uint32_t lcnt[kWorkBins], gbase[kWorkBins];
    lcnt[tid] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + tid;
    uint32_t b = 0, local = 0;
    if (i < n) {
        const uint4 it = items[i];
        b = item_work_bin(it.z, ix.vec_off[it.x + 1] - ix.vec_off[it.x]);
        local = atomicAdd(&lcnt[b], 1u);
    }
    __syncthreads();
    if (lcnt[tid]) gbase[tid] = atomicAdd(&bins[kWorkBins + tid], lcnt[tid]);
    __syncthreads();
    if (i < n) order[start[b] + gbase[b] + local] = i;
}

// The Hamming scan leaf by leaf, for batches in which a leaf is reached by several queries: one block per work item
// (leaf, up to 64 of the queries that reach it; pair_scan_kernel cuts the items; blocks past the last item exit).
// The item's query codes are staged in LDS once; each of the block's two waves then takes 64-vector tiles of the
// leaf, a lane keeping the code of one vector in registers (the planar layout: one coalesced 16-byte load per pair
// of words) while the queries go by, their codes read from LDS at an address that is the same for every lane.
// Every (query, vector) distance goes to the query's row of the spill buffer at the candidate's scan position, where
// select_rows_kernel picks it up. The leaf's code bytes are read once per item instead of once per
// query: compute_hamming_distances (hierarchical.rs:591-614) for all of collect_leaf_candidates' pairs (:618-641),
// grouped by leaf. (Measured on 1000 x 232 leaves of 960-d codes, 94 us: two vectors per lane, two queries per turn
// with explicit counting adds, and items cut in two for a shorter tail were all no faster.)
template <class DistT, int PAIRS, bool TAIL>
__global__ void __launch_bounds__(128)
leaf_hamming_kernel(IndexDev ix, const uint32_t *n_items, const uint4 *items, const uint32_t *order, const uint2 *sorted,
                    const uint64_t *__restrict__ qcodes, DistT *gdist, uint32_t gdist_stride) {
    constexpr uint32_t W = 2 * PAIRS + (TAIL ? 1 : 0), WP = 2 * PAIRS + (TAIL ? 2 : 0); // LDS rows of whole 16-byte pieces
    __shared__ __attribute__((aligned(16))) uint64_t sq[64 * WP];
    __shared__ uint2 spair[64];
    if (blockIdx.x >= *n_items) return;
    CUDDLE_STAMP(10, 0);
    const uint4 it = items[order ? order[blockIdx.x] : blockIdx.x]; // (leaf, first pair, pairs, -); order: heaviest first
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t leaf = it.x, first = it.y, np = it.z;
    if (tid < np) spair[tid] = sorted[first + tid];
    const uint32_t beg = ix.vec_off[leaf], n = ix.vec_off[leaf + 1] - beg;
    const uint4 *planes = ix.codes + ix.code_off[leaf];
    // the wave's first tile of the leaf is requested before the queries are staged: its loads need nothing the
    // staging produces, and a block's time is mostly the dependent trips to memory it makes
    uint4 c[PAIRS > 0 ? PAIRS : 1];
    uint2 ct = make_uint2(0, 0);
    auto request = [&](uint32_t m0) {
        const uint32_t ii = min(m0 + lane, n - 1);
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) c[p] = planes[(uint64_t)p * n + ii];
        if (TAIL) ct = reinterpret_cast<const uint2 *>(planes + (uint64_t)PAIRS * n)[ii];
    };
    if (64 * w < n) request(64 * w);
    __syncthreads();
    CUDDLE_STAMP(10, 1);
    for (uint32_t t = tid; t < np * WP; t += 128) {
        const uint32_t p = t / WP, u = t % WP;
        sq[t] = u < W ? qcodes[(uint64_t)spair[p].x * W + u] : 0ull;
    }
    __syncthreads();
    CUDDLE_STAMP(10, 2);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4 LdsU4;
    LdsU4 *lq = (LdsU4 *)sq;
    for (uint32_t m0 = 64 * w; m0 < n; m0 += 128) {
        const uint32_t i = m0 + lane;
        if (m0 != 64 * w) request(m0);
        // two queries per turn, written out (the row addresses below come from wave-uniform reads, which the compiler
        // will not unroll across); an odd last turn computes its one query twice and stores it once
        for (uint32_t p = 0; p < np; p += 2) {
            const uint32_t p1 = min(p + 1, np - 1);
            LdsU4 *qa = lq + p * (WP / 2), *qb = lq + p1 * (WP / 2);
            // one chain of popcount-accumulates per query (left to itself the compiler counts every 64-bit word apart
            // and adds the counts: five instructions per word instead of four)
            uint32_t ha = 0, hb = 0;
#pragma unroll
            for (int u = 0; u < PAIRS; ++u) {
                const u32x4 va = qa[u], vb = qb[u];
                popc_acc(ha, c[u].x ^ va.x);
                popc_acc(hb, c[u].x ^ vb.x);
                popc_acc(ha, c[u].y ^ va.y);
                popc_acc(hb, c[u].y ^ vb.y);
                popc_acc(ha, c[u].z ^ va.z);
                popc_acc(hb, c[u].z ^ vb.z);
                popc_acc(ha, c[u].w ^ va.w);
                popc_acc(hb, c[u].w ^ vb.w);
            }
            if (TAIL) {
                const u32x4 va = qa[PAIRS], vb = qb[PAIRS];
                popc_acc(ha, ct.x ^ va.x);
                popc_acc(hb, ct.x ^ vb.x);
                popc_acc(ha, ct.y ^ va.y);
                popc_acc(hb, ct.y ^ vb.y);
            }
            // a pair is the same for every lane: its row's address is scalar arithmetic, the lane adds its own element
            const uint2 pa = spair[p], pb = spair[p1];
            DistT *ra = gdist + (uint64_t)__builtin_amdgcn_readfirstlane(pa.x) * gdist_stride + __builtin_amdgcn_readfirstlane(pa.y);
            DistT *rb = gdist + (uint64_t)__builtin_amdgcn_readfirstlane(pb.x) * gdist_stride + __builtin_amdgcn_readfirstlane(pb.y);
            if (i < n) {
                ra[i] = (DistT)ha;
                if (p1 != p) rb[i] = (DistT)hb;
            }
        }
    }
    CUDDLE_STAMP(10, 3);
}

// Measuring aid: the memory side of scan_select_hamming_kernel and nothing else. Same grid, block, LDS size and
// prologue (the block's tables, its tile list), and exactly the scan's loads - the same 16-byte pieces of the same
// leaves, U tiles in flight per wave - but no XOR / popcount, no histogram, no distance stores, no selection: the
// loaded words are folded into one register, which is stored only if it equals a value it practically never has
// (so that the loads stay). Its duration is what the access pattern costs; the scan's excess over it is arithmetic
// and selection.
template <int PAIRS, bool TAIL, int U>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((PAIRS == 1 && !TAIL) ? 8 : 1)))
scan_read_ceiling_kernel(IndexDev ix, const uint32_t *tables, uint32_t tstride, uint32_t lcap, uint32_t tcap, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    ScanLds s = scan_lds_carve(smem, ix.dim, ix.words, lcap, tcap, PAIRS == 1 && !TAIL);
    uint32_t ntiles = 0, q, nl;
    scan_copy_tables(s, tables + (uint64_t)blockIdx.x * 4 * tstride, tstride, lcap, &q, &nl);
    const uint32_t C = scan_build_tiles(s, nl, tcap, &ntiles);
    if (C == 0) return;
    uint32_t acc = 0;
    if (PAIRS == 1 && !TAIL) {
        for (uint32_t t0 = w * U; t0 < ntiles; t0 += nw * U) {
            uint4 c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t t = t0 + u;
                const uint32_t ti = t < ntiles ? s.tinfo[t] : 0u;
                const uint32_t co = s.tcode[t < ntiles ? t : t0];
                c[u] = lane < (ti & 127u) ? ix.codes[(uint64_t)co + lane] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= c[u].x ^ c[u].y ^ c[u].z ^ c[u].w;
        }
    } else {
        for (uint32_t t0 = w * U; t0 < ntiles; t0 += nw * U) {
            uint4 c[U][PAIRS > 0 ? PAIRS : 1];
            uint2 ct[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t t = t0 + u;
                const uint32_t tile = s.tiles[t < ntiles ? t : t0];
                const uint32_t sl = tile >> 12, i = ((tile & 4095u) << 6) + lane;
                const uint32_t base = s.lscan[sl], n = s.lscan[sl + 1] - base;
                const bool act = t < ntiles && i < n;
                const uint4 *planes = ix.codes + s.lcode[sl];
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) c[u][p] = act ? planes[(uint64_t)p * n + i] : make_uint4(0, 0, 0, 0);
                if (TAIL) {
                    const uint2 *tail = reinterpret_cast<const uint2 *>(planes + (uint64_t)PAIRS * n);
                    ct[u] = act ? tail[i] : make_uint2(0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) acc ^= c[u][p].x ^ c[u][p].y ^ c[u][p].z ^ c[u][p].w;
                if (TAIL) acc ^= ct[u].x ^ ct[u].y;
            }
        }
    }
    if (acc == 0x9E3779B9u && sink) sink[q] = acc;
}

// collect_leaf_candidates (hierarchical.rs:618-641) written out in full: every candidate's
// leaf-contiguous position and Hamming distance in scan order. Stage entry point for parity tests.
__global__ void __launch_bounds__(256)
scan_all_hamming_kernel(IndexDev ix, const uint64_t *qcodes, const uint32_t *leaves, uint32_t lstride,
                        const uint32_t *lcount, uint32_t lcap, uint64_t *out_pos, uint32_t *out_key,
                        uint32_t cand_stride, uint32_t *cand_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t W = ix.words;
    ScanLds s = scan_lds_carve(smem, ix.dim, W, lcap, 0);
    const uint32_t nl = min(lcount[q], lcap);
    for (uint32_t i = tid; i < W; i += blockDim.x) s.qw[i] = qcodes[(uint64_t)q * W + i];
    const uint32_t C = scan_load_leaves(ix, s, leaves + (uint64_t)q * lstride, nl, 0, nullptr);
    if (tid == 0) cand_count[q] = C;
    for (uint32_t sl = w; sl < nl; sl += nw) {
        const uint32_t n = s.lscan[sl + 1] - s.lscan[sl], base = s.lscan[sl];
        const uint4 *planes = ix.codes + s.lcode[sl];
        for (uint32_t i = lane; i < n; i += 64) {
            uint32_t h = leaf_hamming(planes, n, i, s.qw, W);
            if (base + i < cand_stride) {
                out_pos[(uint64_t)q * cand_stride + base + i] = s.lbeg[sl] + i;
                out_key[(uint64_t)q * cand_stride + base + i] = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rerank (hierarchical.rs:659-674) and final top-k (:678-691)
// ---------------------------------------------------------------------------------------------

// distance(query, full_vectors.get(idx), metric) for every selected candidate. grid =
// (ceil(sel_stride/64), nq), block 256: the query is staged in LDS once per block, each 4-lane
// group streams one candidate row with 128-bit loads (64 contiguous bytes per group and step).
// order != null: grid = (ceil(sel_stride/64), nq rounded up to a multiple of 8).
#ifndef CUDDLE_RERANK_U
#define CUDDLE_RERANK_U 8
#define CUDDLE_RERANK_W 8
#endif
__global__ void __launch_bounds__(256, CUDDLE_RERANK_W)
rerank_kernel(IndexDev ix, const float *queries, const uint32_t *sel_pos, uint32_t sel_stride,
              const uint32_t *sel_count, float *sel_dist, const float *qnorm, const uint32_t *order, uint32_t nq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);
    // order: a permutation of the queries (scan_tables_kernel's rr_order: neighbours reach the same leaves). Workgroups go
    // to the 8 XCDs in turn by their linear number, and every XCD has an L2 of its own: XCD x takes the x-th eighth of
    // the order, so that the queries that share leaves share an L2
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (order) { // gridDim.y = nq rounded up to a multiple of 8: every XCD has gridDim.y / 8 places, of which the last may be empty
        const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.y >> 3, slot = lin >> 3;
        by = (lin & 7) * per + slot / gridDim.x;
        bx = slot % gridDim.x;
    }
    if (by >= nq) return;
    const uint32_t q = order ? order[by] : by;
    const uint32_t n = sel_count[q];
    if (bx * 64 >= n) return;
    CUDDLE_STAMP_AT(11, 0, blockIdx.y * gridDim.x + blockIdx.x);
    const uint32_t D = ix.dim;
    const float *qg = queries + (uint64_t)q * D;
    for (uint32_t i = threadIdx.x; i < D; i += blockDim.x) qs[i] = qg[i];
    __syncthreads();
    CUDDLE_STAMP_AT(11, 1, blockIdx.y * gridDim.x + blockIdx.x);
    const uint32_t j = bx * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    // a group past the selection does nothing: its four lanes leave together, so the quad exchanges of the groups
    // that stay are among active lanes only (and it reads no row)
    if (j >= n) return;
    const uint32_t pos = sel_pos[(uint64_t)q * sel_stride + j];
    const bool vec = (D & 3) == 0;
    float d = distance_group4_qn_far<CUDDLE_RERANK_U>(qs, ix.vectors + (uint64_t)pos * D, D, ix.metric, sub, vec,
                                         ix.metric == kMetricCosine ? qnorm[q] : 0.f);
    if (sub == 0) sel_dist[(uint64_t)q * sel_stride + j] = d;
    CUDDLE_STAMP_AT(11, 2, blockIdx.y * gridDim.x + blockIdx.x);
}

// finalize_results: ascending by (distance, phase-1 order). For the Hamming path phase-1 order is
// (Hamming distance, scan order) and the selection arrives in scan order, so the sort key is
// distance key (32 bits) | Hamming (16 bits) | index in the selection (14 bits). When
// key_is_rank != 0 the selection already arrives in phase-1 order and the middle field is 0.
// One block per query; dynamic LDS = next_pow2(max selection) u64.
__global__ void __launch_bounds__(256)
topk_kernel(IndexDev ix, const uint32_t *sel_pos, const uint32_t *sel_key, const float *sel_dist,
            uint32_t sel_stride, const uint32_t *sel_count, uint32_t k, uint32_t key_is_rank, uint64_t *out_ids,
            float *out_dist, uint32_t *out_count, const uint64_t *sel_ord, uint64_t *out_order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = sel_count[q];
    const uint32_t npad = next_pow2(max(n, 1u));
    const uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    const float *sd = sel_dist + (uint64_t)q * sel_stride;
    const uint32_t last = max(n, 1u) - 1;
    for (uint32_t i0 = tid; i0 < npad; i0 += 4 * blockDim.x) { // four entries per thread and trip to memory
        float dv[4];
        uint32_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dv[u] = sd[min(i0 + u * blockDim.x, last)];
#pragma unroll
        for (int u = 0; u < 4; ++u) kv[u] = key_is_rank ? 0u : (sk[min(i0 + u * blockDim.x, last)] & 0xFFFFu);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * blockDim.x;
            if (i < npad) skeys[i] = i < n ? (((uint64_t)f32_key(dv[u]) << 30) | ((uint64_t)kv[u] << 14) | i) : ~0ull;
        }
    }
    __syncthreads();
    block_bitonic_sort_u64(skeys, npad);
    const uint32_t nout = min(n, k);
    for (uint32_t i = tid; i < k; i += blockDim.x) {
        if (i < nout) {
            uint32_t j = (uint32_t)(skeys[i] & 0x3FFFu);
            out_ids[(uint64_t)q * k + i] = ix.ids[sp[j]];
            out_dist[(uint64_t)q * k + i] = sd[j];
            if (out_order) out_order[(uint64_t)q * k + i] = sel_ord[(uint64_t)q * sel_stride + j];
        } else {
            out_ids[(uint64_t)q * k + i] = ~0ull;
            out_dist[(uint64_t)q * k + i] = __builtin_inff();
            if (out_order) out_order[(uint64_t)q * k + i] = ~0ull;
        }
    }
    if (tid == 0 && out_count) out_count[q] = nout;
}

// Smallest u64 of the wave, in every lane, by DPP moves (no LDS crossbar): row shifts 1/2/4/8 leave
// each 16-lane row's minimum in its last lane, row_bcast 15 / 31 carry it across the rows into lane
// 63, a readlane broadcasts it. Lanes without a source keep the identity ~0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_min_step_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    return o < v ? o : v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    v = dpp_min_step_u64<0x111, 0xF>(v); // row_shr:1
    v = dpp_min_step_u64<0x112, 0xF>(v); // row_shr:2
    v = dpp_min_step_u64<0x114, 0xF>(v); // row_shr:4
    v = dpp_min_step_u64<0x118, 0xF>(v); // row_shr:8
    v = dpp_min_step_u64<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_min_step_u64<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

// topk_kernel for k <= 64 and selections of at most 64 * KPL candidates: one WAVE per query (one
// query per 64 threads of the block), the keys in registers (KPL per lane), k rounds of "smallest
// key of the wave, its owner retires it". No LDS, no barrier. Same keys, same order.
template <int KPL>
__global__ void __launch_bounds__(256)
topk_wave_kernel(IndexDev ix, const uint32_t *sel_pos, const uint32_t *sel_key, const float *sel_dist,
                 uint32_t sel_stride, const uint32_t *sel_count, uint32_t k, uint32_t key_is_rank, uint64_t *out_ids,
                 float *out_dist, uint32_t *out_count, const uint64_t *sel_ord, uint64_t *out_order, uint32_t nq) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const uint32_t n = min(sel_count[q], 64u * KPL);
    const uint32_t *sp = sel_pos + (uint64_t)q * sel_stride, *sk = sel_key + (uint64_t)q * sel_stride;
    const float *sd = sel_dist + (uint64_t)q * sel_stride;
    uint64_t keys[KPL];
    // every load is issued before the first value is used (loads behind a per-element condition are waited for one by
    // one); slots past the selection re-read its last entry and are masked afterwards
    const uint32_t last = max(n, 1u) - 1;
    float dv[KPL];
    uint32_t kv[KPL];
#pragma unroll
    for (int t = 0; t < KPL; ++t) dv[t] = sd[min((uint32_t)t * 64 + lane, last)];
    if (!key_is_rank) {
#pragma unroll
        for (int t = 0; t < KPL; ++t) kv[t] = sk[min((uint32_t)t * 64 + lane, last)] & 0xFFFFu;
    } else {
#pragma unroll
        for (int t = 0; t < KPL; ++t) kv[t] = 0u;
    }
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
        const uint32_t i = t * 64 + lane;
        keys[t] = i < n ? (((uint64_t)f32_key(dv[t]) << 30) | ((uint64_t)kv[t] << 14) | i) : ~0ull;
    }
    const uint32_t nout = min(n, k); // k <= 64: lane r keeps the r-th winner
    uint64_t mine = ~0ull;
    for (uint32_t r = 0; r < nout; ++r) {
        uint64_t m = keys[0];
#pragma unroll
        for (int t = 1; t < KPL; ++t) m = keys[t] < m ? keys[t] : m;
        const uint64_t wm = wave_min_u64(m);
        if (lane == r) mine = wm;
        if (m == wm) { // keys are distinct (the index is part of them): exactly one lane owns the minimum
#pragma unroll
            for (int t = 0; t < KPL; ++t) keys[t] = keys[t] == wm ? ~0ull : keys[t];
        }
    }
    // all winners gathered at once: one chain of dependent loads for the wave, not one per round
    for (uint32_t i = lane; i < k; i += 64) {
        if (i < nout) {
            const uint32_t j = (uint32_t)(mine & 0x3FFFu);
            out_ids[(uint64_t)q * k + i] = ix.ids[sp[j]];
            out_dist[(uint64_t)q * k + i] = sd[j];
            if (out_order) out_order[(uint64_t)q * k + i] = sel_ord[(uint64_t)q * sel_stride + j];
        } else {
            out_ids[(uint64_t)q * k + i] = ~0ull;
            out_dist[(uint64_t)q * k + i] = __builtin_inff();
            if (out_order) out_order[(uint64_t)q * k + i] = ~0ull;
        }
    }
    if (lane == 0 && out_count) out_count[q] = nout;
}

// Phase-1 selection listed in phase-1 order (stage output for parity tests): sorts the scan-order
// selection by (key, scan order) and gathers pos/key/dist. One block per query.
__global__ void __launch_bounds__(256)
selection_order_kernel(const uint32_t *sel_pos, const uint32_t *sel_key, const float *sel_dist, uint32_t sel_stride,
                       const uint32_t *sel_count, uint32_t key_is_f32, uint64_t *out_pos, uint32_t *out_key,
                       float *out_dist, uint32_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = sel_count[q];
    const uint32_t npad = next_pow2(max(n, 1u));
    for (uint32_t i = tid; i < npad; i += blockDim.x) {
        uint64_t key = ~0ull;
        if (i < n) {
            uint32_t kv = sel_key[(uint64_t)q * sel_stride + i];
            if (key_is_f32) kv = f32_key(__uint_as_float(kv));
            key = ((uint64_t)kv << 32) | i;
        }
        skeys[i] = key;
    }
    __syncthreads();
    block_bitonic_sort_u64(skeys, npad);
    for (uint32_t i = tid; i < n && i < out_stride; i += blockDim.x) {
        uint32_t j = (uint32_t)skeys[i];
        out_pos[(uint64_t)q * out_stride + i] = sel_pos[(uint64_t)q * sel_stride + j];
        out_key[(uint64_t)q * out_stride + i] = sel_key[(uint64_t)q * sel_stride + j];
        if (out_dist) out_dist[(uint64_t)q * out_stride + i] = sel_dist[(uint64_t)q * sel_stride + j];
    }
}

} // namespace cuddle
