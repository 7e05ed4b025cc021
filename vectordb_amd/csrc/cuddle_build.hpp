// cuddle_build.hpp — index-build side of the path: exact nearest-centroid assignment, the inner
// step of k-means (reference: src/clustering/kmeans.rs:136-160, KMeans::assign; also
// nearest_centroid, :229-236); the centroid update (:163-215); the distance step of k-means++
// seeding (:55-64); RaBitQ encoding of the stored vectors (src/quantization/rabitq.rs:114-171).
//
// The reference evaluates distance(v, c, metric) for every (vector, centroid) pair and keeps the
// first minimum (Iterator::min_by returns the first of equal minima). Here the n x k distance matrix
// is never materialised: the matrix-core contraction of cuddle_route.hpp produces approximate
// distances tile by tile, each block reduces its tiles (a segment of at most 256 centroids) to the
// three smallest per vector, and a second kernel evaluates the exact distance only where the error
// bound of the contraction says the minimum could be: normally one centroid per vector. Where a
// segment holds three or more such candidates (near-ties, duplicate centroids) the vector is compared
// exactly with every centroid of that segment. The result is the reference's, bit for bit.
#pragma once

#include "cuddle_route.hpp"

namespace cuddle {

constexpr size_t kAssignKeyTileBytes = (size_t)kGemmBM * 65 * 4; // approximate-distance keys of one 128 x 64 tile

template <int BK>
constexpr size_t assign_tile_lds_bytes() { // the contraction's operands, then the tile's keys, share this space
    return route_gemm_lds_bytes<1, BK>() > kAssignKeyTileBytes ? route_gemm_lds_bytes<1, BK>() : kAssignKeyTileBytes;
}
template <int BK>
constexpr size_t assign_gemm_lds_bytes() { return assign_tile_lds_bytes<BK>() + kGemmBM * sizeof(float); }

// The three smallest keys of a run, with the columns of the first two; equal keys count
// separately (m2 == m1 when the minimum occurs twice).
struct Best3 {
    uint32_t m1, a1, m2, a2, m3;
    __device__ __forceinline__ void add(uint32_t key, uint32_t col) {
        if (key < m2) {
            m3 = m2;
            if (key < m1) {
                m2 = m1;
                a2 = a1;
                m1 = key;
                a1 = col;
            } else {
                m2 = key;
                a2 = col;
            }
        } else if (key < m3) {
            m3 = key;
        }
    }
};

constexpr uint32_t kAssignMaxTilesPerSeg = 4; // a segment summary covers at most 256 centroids

// Approximate nearest centroids. A: vectors minus mu, split form [M][kp]; B: centroids minus mu,
// split form [N][kp]. Block (row tile by, column segment seg) walks the segment's column tiles of 64
// centroids; after each contraction the 128 x 64 keys (monotone u32 image of d~) go through LDS so
// that thread (row, half) sees 32 of its row's keys and folds them into its running Best3.
// best[(row * cs + seg) * 2] = (m1, a1, m2, a2), [.. + 1] = (m3, 0, 0, 0).
// grid = gy * cs, block 256, LDS assign_gemm_lds_bytes<BK>().
template <int BK>
__global__ void __launch_bounds__(256, 3)
assign_gemm_kernel(const u32x4 *A, const float *Asq, const u32x4 *B, const float *Bsq, uint32_t M, uint32_t N, uint32_t kp,
                   uint32_t metric, uint4 *best, uint32_t cs, uint32_t tiles_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *kt = reinterpret_cast<uint32_t *>(smem); // [128][65]
    const uint32_t by = blockIdx.x / cs, seg = blockIdx.x % cs;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t wr = w >> 1, wc = w & 1, r = lane & 31, h = lane >> 5;
    const uint32_t gx = (N + 63) / 64;
    const uint32_t t0 = seg * tiles_per_seg, t1 = min(t0 + tiles_per_seg, gx);
    const uint32_t row = tid >> 1, half = tid & 1;
    Best3 b{0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu};
    float *asq = reinterpret_cast<float *>(smem + assign_tile_lds_bytes<BK>()); // [128] squared norms of the block's rows
    if (tid < kGemmBM) asq[tid] = Asq[min(by * kGemmBM + tid, M - 1)];
    // (visible to every thread after the barriers inside the first contraction)
    for (uint32_t bx = t0; bx < t1; ++bx) {
        f32x16 acc[2][1], cor[2][1];
        split_gemm_tile<1, BK>(A, B, M, N, kp, by, bx, smem, acc, cor);
        __syncthreads(); // the operands in LDS have been read: reuse the space for the keys
        const uint32_t gn = bx * 64 + wc * 32 + r;
        const float bs = Bsq[min(gn, N - 1)];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t lrow = wr * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float d = approx_distance(metric, acc[mt][0][e] + cor[mt][0][e], asq[lrow], bs);
                kt[lrow * 65 + wc * 32 + r] = gn < N ? f32_key(d) : 0xFFFFFFFFu;
            }
        __syncthreads();
        const uint32_t *kr = kt + row * 65 + half * 32;
#pragma unroll 8
        for (uint32_t c = 0; c < 32; ++c) b.add(kr[c], bx * 64 + half * 32 + c);
        // the next tile's first LDS store comes after the barrier that opens its K loop
    }
    // fold the other half of the row in (columns beyond N carry the key 0xFFFFFFFF and never displace a real one)
    const uint32_t pm1 = (uint32_t)__shfl_xor((int)b.m1, 1, 64), pa1 = (uint32_t)__shfl_xor((int)b.a1, 1, 64),
                   pm2 = (uint32_t)__shfl_xor((int)b.m2, 1, 64), pa2 = (uint32_t)__shfl_xor((int)b.a2, 1, 64),
                   pm3 = (uint32_t)__shfl_xor((int)b.m3, 1, 64);
    b.add(pm1, pa1);
    b.add(pm2, pa2);
    b.m3 = min(b.m3, pm3);
    const uint32_t gm = by * kGemmBM + row;
    if (half == 0 && gm < M) {
        uint4 *o = best + ((uint64_t)gm * cs + seg) * 2;
        o[0] = make_uint4(b.m1, b.a1, b.m2, b.a2);
        o[1] = make_uint4(b.m3, 0u, 0u, 0u);
    }
}

// Exact decision per vector from the segment summaries. With g = the smallest approximate distance
// of the row and e the bound on |d~ - d|, the true nearest centroid has d~ <= g + 2e =: t. Per segment:
// m1 > t: nothing there; m3 > t: its candidates are a1 and, if m2 <= t, a2; m3 <= t: the summary may
// hide more, so every centroid of the segment is evaluated (counted in `fallbacks`). Exact distances
// are distance(v, c, metric) in the reference's accumulation order; ties go to the lower index.
// One 4-lane group per vector; grid = ceil(M / 64), block 256. vectors: the original f32 rows of this
// chunk.
__global__ void __launch_bounds__(256)
assign_finish_kernel(const float *vectors, const float *centroids, uint32_t M, uint32_t N, uint32_t dim, uint32_t metric,
                     const uint4 *best, uint32_t cs, uint32_t tiles_per_seg, const float *Asq, float max_csq,
                     uint32_t *assign, float *dist, uint32_t *fallbacks) {
    const uint32_t row = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    const bool act = row < M;
    const uint32_t rr = act ? row : M - 1; // inactive groups shadow the last row: the group reductions need all lanes
    const float *v = vectors + (uint64_t)rr * dim;
    const uint4 *bb = best + (uint64_t)rr * cs * 2;
    const bool vec = (dim & 3) == 0;
    uint32_t g = 0xFFFFFFFFu;
    for (uint32_t s = 0; s < cs; ++s) g = min(g, bb[2 * s].x);
    const float e = route_error_bound(metric, dim, Asq[rr], max_csq);
    const float lim = f32_from_key(g) + 2.0f * e;
    const uint32_t thr = f32_key(lim + fabsf(lim) * 2.4e-7f); // allowance for the rounding of the sum above
    uint32_t best_i = 0xFFFFFFFFu;
    float best_d = 0.f;
    auto consider = [&](uint32_t c) {
        const float d = distance_group4(v, centroids + (uint64_t)c * dim, dim, metric, sub, vec);
        if (best_i == 0xFFFFFFFFu || d < best_d || (d == best_d && c < best_i)) {
            best_d = d;
            best_i = c;
        }
    };
    uint32_t full = 0;
    for (uint32_t s = 0; s < cs; ++s) {
        const uint4 b0 = bb[2 * s];
        if (b0.x > thr || b0.x == 0xFFFFFFFFu) continue;
        if (bb[2 * s + 1].x <= thr) {
            const uint32_t c0 = s * tiles_per_seg * 64, c1 = min(c0 + tiles_per_seg * 64, N);
            for (uint32_t c = c0; c < c1; ++c) consider(c);
            ++full;
        } else {
            consider(b0.y);
            if (b0.z <= thr) consider(b0.w);
        }
    }
    if (act && sub == 0) {
        assign[row] = best_i;
        dist[row] = best_d;
        if (full) atomicAdd(fallbacks, full);
    }
}

// ---------------------------------------------------------------------------------------------
// centroid update (KMeans::update_centroids, kmeans.rs:163-215)
// ---------------------------------------------------------------------------------------------
// The reference adds a cluster's vectors up per chunk of 2048 consecutive vectors, one sequential f32
// accumulation per (chunk, cluster, component), and merges the chunks' sums in chunk order; here
// kmeans_partial_kernel produces the per-chunk sums in exactly that order and kmeans_fold_kernel merges
// them from the left (a chunk without vectors of a cluster contributes +0.0, which changes nothing).
constexpr uint32_t kUpdateChunk = 2048;      // vectors per partial sum (kmeans.rs:170)
constexpr uint32_t kUpdateAccFloats = 11264; // accumulator floats a block keeps in LDS (44 KB + the 8 KB list: three blocks per CU)

__global__ void __launch_bounds__(256)
kmeans_count_kernel(const uint32_t *assign, uint64_t n, uint32_t k, uint32_t *counts) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = assign[i];
    if (c < k) atomicAdd(&counts[c], 1u);
}

// Block (chunk, cluster range r) sums the chunk's vectors of clusters [r * kr, (r + 1) * kr) in vector
// order: thread t owns components t, t + blockDim, ...; the accumulators of the range live in LDS.
// part: [chunks of this launch][k][dim]. Vectors whose cluster id is not below k are ignored.
// grid = (chunks, ceil(k / kr)), block = 64, 128 or 256 threads, dynamic LDS = kr * dim * 4 + 2048 * 4 bytes.
__global__ void __launch_bounds__(256)
kmeans_partial_kernel(const float *vectors, const uint32_t *assign, uint64_t n, uint32_t dim, uint32_t k, uint32_t kr,
                      uint64_t first_chunk, float *part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                                  // [kr][dim]
    uint32_t *list = reinterpret_cast<uint32_t *>(acc + (size_t)kr * dim);        // [2048] (position in chunk) << 16 | local cluster
    __shared__ uint32_t wsum[4], s_n;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint64_t base = (first_chunk + blockIdx.x) * kUpdateChunk;
    const uint32_t m = (uint32_t)min((uint64_t)kUpdateChunk, n - base);
    const uint32_t c0 = blockIdx.y * kr, c1 = min(c0 + kr, k);
    for (uint32_t i = tid; i < (c1 - c0) * dim; i += nt) acc[i] = 0.f;
    // the chunk's vectors of this range, in vector order: thread t looks at positions [t * per, (t + 1) * per)
    const uint32_t per = kUpdateChunk / nt;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t p = tid * per + j;
        const uint32_t c = p < m ? assign[base + p] : 0xFFFFFFFFu;
        mine += (c >= c0 && c < c1) ? 1u : 0u;
    }
    const uint32_t incl = wave_inclusive_sum(mine);
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - mine;
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += wsum[w];
    if (tid == nt - 1) s_n = off + mine;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t p = tid * per + j;
        const uint32_t c = p < m ? assign[base + p] : 0xFFFFFFFFu;
        if (c >= c0 && c < c1) list[off++] = (p << 16) | (c - c0);
    }
    __syncthreads();
    const uint32_t cnt = s_n;
    const float *vb = vectors + base * dim;
    for (uint32_t i = tid; i < dim; i += nt) {
        uint32_t j = 0;
        for (; j + 8 <= cnt; j += 8) { // eight rows in flight, then the eight additions in order
            uint32_t e[8];
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = list[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = vb[(uint64_t)(e[u] >> 16) * dim + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[(e[u] & 0xFFFFu) * dim + i] += x[u];
        }
        for (; j < cnt; ++j) {
            const uint32_t e = list[j];
            acc[(e & 0xFFFFu) * dim + i] += vb[(uint64_t)(e >> 16) * dim + i];
        }
    }
    __syncthreads();
    float *out = part + ((uint64_t)blockIdx.x * k + c0) * dim;
    for (uint32_t i = tid; i < (c1 - c0) * dim; i += nt) out[i] = acc[i];
}

// total[c][i] += part[0][c][i], then part[1][c][i], ... (left to right); with `last`, the mean
// total / (count as f32) goes to the centroid rows of the clusters that have vectors (kmeans.rs:204-215).
__global__ void __launch_bounds__(256)
kmeans_fold_kernel(const float *part, uint32_t chunks, uint32_t k, uint32_t dim, float *total, const uint32_t *counts,
                   float *centroids, uint32_t last) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, kd = (uint64_t)k * dim;
    if (g >= kd) return;
    float t = total[g];
    uint32_t ch = 0;
    for (; ch + 8 <= chunks; ch += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = part[(uint64_t)(ch + u) * kd + g];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += x[u];
    }
    for (; ch < chunks; ++ch) t += part[(uint64_t)ch * kd + g];
    total[g] = t;
    if (last) {
        const uint32_t cnt = counts[g / dim];
        if (cnt) centroids[g] = t / (float)cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// k-means++ seeding, the parallel half of a round (KMeans::init_plusplus, kmeans.rs:55-67)
// ---------------------------------------------------------------------------------------------
// mind[v] = min(mind[v], distance(v, newest centroid)) (first round: the distance itself; the minimum
// over the centroids chosen so far is the same value whichever order they are visited in), and
// w[v] = mind[v] * mind[v], the weight the reference samples with. One 4-lane group per vector.
__global__ void __launch_bounds__(256)
kmeanspp_round_kernel(const float *vectors, uint64_t n, uint32_t dim, uint32_t metric, const float *centroid, float *mind,
                      float *w, uint32_t first_round) {
    const uint64_t v = (uint64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const uint32_t sub = threadIdx.x & 3;
    const bool act = v < n;
    const uint64_t vv = act ? v : n - 1; // idle groups shadow the last vector: the group reductions need all lanes
    const float d = distance_group4(vectors + vv * dim, centroid, dim, metric, sub, (dim & 3) == 0);
    if (act && sub == 0) {
        const float m = first_round ? d : fminf(mind[v], d);
        mind[v] = m;
        w[v] = m * m;
    }
}

// ---------------------------------------------------------------------------------------------
// RaBitQ encoding (RaBitQQuantizer::quantize, rabitq.rs:114-171)
// ---------------------------------------------------------------------------------------------
// Per vector: norm = sqrt(sequential sum of squares) (:118); zero vectors (norm < 1e-10): code 0, norm 0,
// <o_bar, o> = 1 (:121-127); otherwise o = v / norm, rotated_i = sum_j rot_t[i][j] * o[j] sequentially over
// j (:138-140), bit i of word i / 32 = rotated_i >= 0 (:143-150), and <o_bar, o> = the sequential sum over i
// of sign_i * rotated_i = |rotated_i| (:157-164). rot: the rotation in the layout transpose_square_kernel
// makes (rot[j][i] = rot_t[i][j]), so that consecutive threads (consecutive i) read consecutive addresses.
// A block encodes VB vectors: each rotation element is read once for all of them.
// grid = ceil(n / VB), block 256, dynamic LDS = 2 * VB * dim * 4 bytes.
template <int VB>
__global__ void __launch_bounds__(256)
rabitq_encode_kernel(const float *vectors, const float *rot, uint64_t n, uint32_t dim, uint32_t *codes, float *norms,
                     float *ip_oo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *o = reinterpret_cast<float *>(smem); // [dim][VB]: component j of the block's normalised vectors side by side
    float *mag = o + (size_t)VB * dim;          // [VB][dim]: |rotated|
    __shared__ float s_norm[VB];
    const uint64_t v0 = (uint64_t)blockIdx.x * VB;
    const uint32_t tid = threadIdx.x, words = (dim + 31) / 32;
    if (tid < VB) {
        const uint64_t v = v0 + tid;
        float sq = 0.f;
        if (v < n)
            for (uint32_t j = 0; j < dim; ++j) {
                const float x = vectors[v * dim + j];
                sq += x * x;
            }
        s_norm[tid] = sqrtf(sq);
    }
    __syncthreads();
    for (uint32_t t = tid; t < VB * dim; t += 256) {
        const uint32_t b = t / dim, j = t - b * dim;
        const uint64_t v = v0 + b;
        o[j * VB + b] = (v < n && s_norm[b] >= 1e-10f) ? vectors[v * dim + j] / s_norm[b] : 0.f;
    }
    __syncthreads();
    const uint32_t dim64 = (dim + 63) & ~63u; // whole waves walk the output rows, so that the ballots below are complete
    for (uint32_t i = tid; i < dim64; i += 256) {
        float acc[VB];
#pragma unroll
        for (int b = 0; b < VB; ++b) acc[b] = 0.f;
        if (i < dim)
            for (uint32_t j = 0; j < dim; ++j) {
                const float r = rot[(uint64_t)j * dim + i];
#pragma unroll
                for (int b = 0; b < VB; ++b) acc[b] += r * o[j * VB + b];
            }
#pragma unroll
        for (int b = 0; b < VB; ++b) {
            const uint64_t bits = __ballot(i < dim && acc[b] >= 0.0f);
            const uint64_t v = v0 + b;
            if ((tid & 31) == 0 && v < n && (i >> 5) < words)
                codes[v * words + (i >> 5)] = s_norm[b] >= 1e-10f ? (uint32_t)(bits >> (tid & 32)) : 0u;
            if (i < dim) mag[b * dim + i] = fabsf(acc[b]);
        }
    }
    __syncthreads();
    if (tid < VB) {
        const uint64_t v = v0 + tid;
        if (v < n) {
            float ip = 0.f;
            for (uint32_t i = 0; i < dim; ++i) ip += mag[tid * dim + i];
            const bool zero = s_norm[tid] < 1e-10f;
            norms[v] = zero ? 0.f : s_norm[tid];
            ip_oo[v] = zero ? 1.f : ip;
        }
    }
}

} // namespace cuddle
