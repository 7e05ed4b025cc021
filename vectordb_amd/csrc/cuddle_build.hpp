// cuddle_build.hpp — index-build side of the path: exact nearest-centroid assignment, the inner
// step of k-means (reference: src/clustering/kmeans.rs:136-160, KMeans::assign; also
// nearest_centroid, :229-236).
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

} // namespace cuddle
