/*
 * cuddle_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the reference's batched-search arithmetic (LLay/vectordb,
 * Rust, aarch64-NEON only; it cannot be compiled in this environment: no Rust toolchain and
 * unconditional `use std::arch::aarch64::*`). Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this file's shared object, and only as the checker
 * or the timed CPU baseline — the product library (libcuddle_hip.so) never links or calls it.
 *
 * Pinning: the reference ships no golden vectors for this path; the restatement is pinned by
 * every analytic known-answer test the reference's own unit tests hold (tests/test_oracle_kats.py
 * lists them with file:line) and by brute-force ground-truth recall.
 *
 * Each function cites the reference lines it restates. Floating point: compiled with
 * -ffp-contract=off; a fused multiply-add appears exactly where the reference uses
 * vfmaq_f32 (written fmaf), a separately rounded multiply where it uses vmulq_f32 or scalar
 * Rust arithmetic (rustc never contracts a*b+c).
 * vaddvq_f32(v) is the pairwise reduction (v0+v1)+(v2+v3) (FADDP twice).
 *
 * Tie rule (the reference leaves ties to select_nth_unstable): phase-1 order is
 * (code distance asc, scan position asc); routing order is (distance asc, frontier
 * position asc); final order is (distance asc, phase-1 order asc) — the order the
 * reference's stable sorts give in search_with_stats (hierarchical.rs:888-894, 930-932) and
 * in the small-input branches of search (:646-648, :679-681).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/cuddle_hip.h"

#define ORC_OK 0
#define ORC_ERR_BAD_ARG 1
#define ORC_ERR_BAD_DIM 2
#define ORC_ERR_OOM 5
#define ORC_ERR_NAN 100 /* the reference would panic: partial_cmp().unwrap() on NaN */

/* ---------------------------------------------------------------------------------------- */
/* src/distance/simd_neon.rs                                                                 */
/* ---------------------------------------------------------------------------------------- */

/* dot_product_neon, simd_neon.rs:13-70: 4 NEON accumulators x 4 lanes = 16 partial sums over
 * elements i = j (mod 16), combined (sum0+sum1)+(sum2+sum3) lane-wise, then vaddvq; then
 * 4-element chunks with an unfused multiply and a pairwise add; then a scalar tail. */
float orc_dot(const float *a, const float *b, size_t n) {
    float s[16];
    size_t i = 0;
    for (int j = 0; j < 16; ++j) s[j] = 0.0f;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j) s[j] = fmaf(a[i + j], b[i + j], s[j]);
    float u[4];
    for (int l = 0; l < 4; ++l) {
        float t0 = s[l] + s[4 + l];
        float t2 = s[8 + l] + s[12 + l];
        u[l] = t0 + t2;
    }
    float r = (u[0] + u[1]) + (u[2] + u[3]);
    for (; i + 4 <= n; i += 4) {
        float p0 = a[i] * b[i], p1 = a[i + 1] * b[i + 1];
        float p2 = a[i + 2] * b[i + 2], p3 = a[i + 3] * b[i + 3];
        r += (p0 + p1) + (p2 + p3);
    }
    for (; i < n; ++i) r += a[i] * b[i];
    return r;
}

/* l2_squared_neon, simd_neon.rs:77-140: same structure on diff = a - b. */
float orc_l2(const float *a, const float *b, size_t n) {
    float s[16];
    size_t i = 0;
    for (int j = 0; j < 16; ++j) s[j] = 0.0f;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j) {
            float d = a[i + j] - b[i + j];
            s[j] = fmaf(d, d, s[j]);
        }
    float u[4];
    for (int l = 0; l < 4; ++l) {
        float t0 = s[l] + s[4 + l];
        float t2 = s[8 + l] + s[12 + l];
        u[l] = t0 + t2;
    }
    float r = (u[0] + u[1]) + (u[2] + u[3]);
    for (; i + 4 <= n; i += 4) {
        float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1];
        float d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
        r += (q0 + q1) + (q2 + q3);
    }
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        r += d * d;
    }
    return r;
}

/* l2_squared_batch4_neon, simd_neon.rs:160-227, for ONE of its four targets (the four are
 * independent): one NEON accumulator (4 lanes, elements i = l (mod 4)), vaddvq, scalar tail. */
static float orc_l2_batch4_one(const float *q, const float *t, size_t dim) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    size_t i = 0;
    for (; i + 4 <= dim; i += 4)
        for (int l = 0; l < 4; ++l) {
            float d = q[i + l] - t[i + l];
            acc[l] = fmaf(d, d, acc[l]);
        }
    float r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    for (; i < dim; ++i) {
        float d = q[i] - t[i];
        r += d * d;
    }
    return r;
}

/* l2_squared_batch, simd_neon.rs:233-257: groups of 4 targets through batch4, the last
 * (n mod 4) targets through l2_squared. targets = nt row pointers. */
void orc_l2_batch(const float *q, const float *const *targets, size_t nt, size_t dim, float *out) {
    size_t i = 0;
    for (; i + 4 <= nt; i += 4)
        for (int j = 0; j < 4; ++j) out[i + j] = orc_l2_batch4_one(q, targets[i + j], dim);
    for (; i < nt; ++i) out[i] = orc_l2(q, targets[i], dim);
}

/* Row-matrix convenience for the tests: targets is [nt][dim]. */
void orc_l2_batch_rows(const float *q, const float *targets, size_t nt, size_t dim, float *out) {
    size_t i = 0;
    for (; i + 4 <= nt; i += 4)
        for (int j = 0; j < 4; ++j) out[i + j] = orc_l2_batch4_one(q, targets + (i + j) * dim, dim);
    for (; i < nt; ++i) out[i] = orc_l2(q, targets + i * dim, dim);
}

/* distance(), src/distance/mod.rs:23-34. */
float orc_distance(const float *a, const float *b, size_t n, int metric) {
    if (metric == CUDDLE_METRIC_L2) return orc_l2(a, b, n);
    if (metric == CUDDLE_METRIC_DOT) return -orc_dot(a, b, n);
    float dot = orc_dot(a, b, n);
    float na = sqrtf(orc_dot(a, a, n));
    float nb = sqrtf(orc_dot(b, b, n));
    return 1.0f - (dot / (na * nb));
}

void orc_distance_pairs(const float *a, const float *b, uint64_t n, uint32_t dim, int metric, float *out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = orc_distance(a + i * dim, b + i * dim, dim, metric);
}

/* KMeans::assign, src/clustering/kmeans.rs:136-160 (nearest_centroid, :229-236, is the same per
 * vector): distance(v, c, metric) against every centroid in order; Iterator::min_by keeps the first
 * of equal minima. */
void orc_kmeans_assign(const float *vectors, uint64_t n, const float *centroids, uint32_t k, uint32_t dim,
                       int metric, uint32_t *assign, float *dist) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint32_t best = 0;
        float bd = orc_distance(vectors + (uint64_t)i * dim, centroids, dim, metric);
        for (uint32_t c = 1; c < k; ++c) {
            float d = orc_distance(vectors + (uint64_t)i * dim, centroids + (uint64_t)c * dim, dim, metric);
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        assign[i] = best;
        dist[i] = bd;
    }
}

/* KMeans::update_centroids, src/clustering/kmeans.rs:163-215. The reference sums each cluster's
 * vectors per chunk of 2048 consecutive vectors (one sequential f32 accumulation per chunk, cluster
 * and component, :178-183) and merges the chunks' partial sums with rayon's reduce (:187-199), which
 * keeps the chunks in order but leaves the grouping of the additions to the scheduler; this
 * restatement merges them from the left, one of the groupings the reference can produce. A chunk in
 * which a cluster has no vector contributes +0.0, which changes nothing. mean = sum / (count as f32)
 * (:209-213); the row of a cluster without vectors is left as it was (the reference replaces it with
 * a randomly drawn vector, :217-223: the caller's business here). counts[c] = vectors of cluster c. */
void orc_kmeans_update(const float *vectors, uint64_t n, uint32_t dim, const uint32_t *assign, uint32_t k,
                       float *centroids, uint32_t *counts) {
    float *total = (float *)calloc((size_t)k * dim, sizeof(float));
    float *part = (float *)malloc((size_t)k * dim * sizeof(float));
    unsigned char *touched = (unsigned char *)malloc(k);
    for (uint32_t c = 0; c < k; ++c) counts[c] = 0;
    for (uint64_t base = 0; base < n; base += 2048) {
        const uint64_t end = base + 2048 < n ? base + 2048 : n;
        memset(touched, 0, k);
        for (uint64_t v = base; v < end; ++v) {
            const uint32_t c = assign[v];
            float *p = part + (size_t)c * dim;
            if (!touched[c]) {
                touched[c] = 1;
                for (uint32_t i = 0; i < dim; ++i) p[i] = 0.0f;
            }
            const float *x = vectors + v * dim;
            for (uint32_t i = 0; i < dim; ++i) p[i] += x[i];
            counts[c] += 1;
        }
        for (uint32_t c = 0; c < k; ++c)
            if (touched[c]) {
                float *t = total + (size_t)c * dim;
                const float *p = part + (size_t)c * dim;
                for (uint32_t i = 0; i < dim; ++i) t[i] += p[i];
            }
    }
    for (uint32_t c = 0; c < k; ++c)
        if (counts[c]) {
            const float cf = (float)counts[c];
            for (uint32_t i = 0; i < dim; ++i) centroids[(size_t)c * dim + i] = total[(size_t)c * dim + i] / cf;
        }
    free(total);
    free(part);
    free(touched);
}

/* KMeans::init_plusplus, src/clustering/kmeans.rs:36-88, with the reference's random draws handed in:
 * `first` = the index vectors.choose() picked (:49), uniforms[j - 1] = the value of rand::random::<f32>()
 * in round j (:68). Per round: every vector's distance to its nearest centroid so far (min over
 * distance(v, c, metric), :55-64), total = the sequential f32 sum of the squares (:67), target = u * total,
 * then the first index at which the running target - d*d drops to <= 0 (:70-77; index 0 when it never
 * does). chosen[j] = index of centroid j; centroids [k][dim] receives the rows. */
void orc_kmeanspp_init(const float *vectors, uint64_t n, uint32_t dim, uint32_t k, int metric, uint64_t first,
                       const float *uniforms, float *centroids, uint64_t *chosen) {
    float *mind = (float *)malloc(n * sizeof(float));
    chosen[0] = first;
    memcpy(centroids, vectors + first * dim, dim * sizeof(float));
    for (uint32_t j = 1; j < k; ++j) {
        const float *cn = centroids + (size_t)(j - 1) * dim;
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < (int64_t)n; ++v) {
            const float d = orc_distance(vectors + (uint64_t)v * dim, cn, dim, metric);
            if (j == 1 || d < mind[v]) mind[v] = d;
        }
        float total = 0.0f;
        for (uint64_t v = 0; v < n; ++v) total += mind[v] * mind[v];
        float target = uniforms[j - 1] * total;
        uint64_t idx = 0;
        for (uint64_t v = 0; v < n; ++v) {
            target -= mind[v] * mind[v];
            if (target <= 0.0f) {
                idx = v;
                break;
            }
        }
        chosen[j] = idx;
        memcpy(centroids + (size_t)j * dim, vectors + idx * dim, dim * sizeof(float));
    }
    free(mind);
}

/* ---------------------------------------------------------------------------------------- */
/* src/quantization/binary.rs                                                                */
/* ---------------------------------------------------------------------------------------- */

/* BinaryQuantizer::quantize, binary.rs:144-159: strict >, word i/64, bit i%64. */
void orc_binary_quantize(const float *v, uint64_t n, uint32_t dim, float threshold, uint64_t *codes) {
    uint32_t words = (dim + 63) / 64;
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t *c = codes + r * words;
        for (uint32_t w = 0; w < words; ++w) c[w] = 0;
        const float *x = v + r * (uint64_t)dim;
        for (uint32_t i = 0; i < dim; ++i)
            if (x[i] > threshold) c[i / 64] |= (uint64_t)1 << (i % 64);
    }
}

/* BinaryQuantizer::from_vectors, binary.rs:122-141: sequential f32 fold over all values in
 * row order, divided by (count as f32). */
float orc_binary_threshold(const float *v, uint64_t n, uint32_t dim) {
    float total = 0.0f;
    uint64_t count = n * (uint64_t)dim;
    for (uint64_t i = 0; i < count; ++i) total += v[i];
    return total / (float)count;
}

/* BinaryVector::hamming_distance, binary.rs:42-54 (the NEON twin :57-95 is the same integer). */
uint32_t orc_hamming(const uint64_t *a, const uint64_t *b, uint32_t words) {
    uint32_t d = 0;
    for (uint32_t w = 0; w < words; ++w) d += (uint32_t)__builtin_popcountll(a[w] ^ b[w]);
    return d;
}

void orc_hamming_pairs(const uint64_t *a, const uint64_t *b, uint64_t n, uint32_t words, uint32_t *out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = orc_hamming(a + i * words, b + i * words, words);
}

/* ---------------------------------------------------------------------------------------- */
/* src/quantization/rabitq.rs                                                                */
/* ---------------------------------------------------------------------------------------- */

/* Sequential f32 sum of x*x then sqrt: rabitq.rs:118, :406. */
static float orc_norm_seq(const float *x, uint32_t dim) {
    float s = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) s += x[i] * x[i];
    return sqrtf(s);
}

/* RaBitQQuantizer::quantize, rabitq.rs:114-171 (index-build side; produces fixtures). */
void orc_rabitq_quantize(const float *rot_t, uint32_t dim, const float *v, uint64_t n, uint32_t *codes,
                         float *norms, float *ip_oo) {
    uint32_t words = (dim + 31) / 32;
    float *normalized = (float *)malloc(sizeof(float) * dim);
    for (uint64_t r = 0; r < n; ++r) {
        const float *x = v + r * (uint64_t)dim;
        uint32_t *code = codes + r * words;
        for (uint32_t w = 0; w < words; ++w) code[w] = 0;
        float norm = orc_norm_seq(x, dim);
        if (norm < 1e-10f) { /* :121-127 */
            norms[r] = 0.0f;
            ip_oo[r] = 1.0f;
            continue;
        }
        for (uint32_t j = 0; j < dim; ++j) normalized[j] = x[j] / norm;
        float ip = 0.0f;
        for (uint32_t i = 0; i < dim; ++i) {
            const float *row = rot_t + (uint64_t)i * dim;
            float rotated = 0.0f;
            for (uint32_t j = 0; j < dim; ++j) rotated += row[j] * normalized[j];
            float qv = rotated >= 0.0f ? 1.0f : -1.0f;
            if (rotated >= 0.0f) code[i / 32] |= 1u << (i % 32);
            ip += qv * rotated; /* :157-164, same rotated value recomputed */
        }
        norms[r] = norm;
        ip_oo[r] = ip;
    }
    free(normalized);
}

/* rotate_query, rabitq.rs:191-201. */
void orc_rabitq_rotate(const float *rot_t, uint32_t dim, const float *qn, float *out) {
    for (uint32_t i = 0; i < dim; ++i) {
        const float *row = rot_t + (uint64_t)i * dim;
        float s = 0.0f;
        for (uint32_t j = 0; j < dim; ++j) s += row[j] * qn[j];
        out[i] = s;
    }
}

/* compute_inner_product_fast, rabitq.rs:298-321. */
static float orc_rabitq_ip_fast(const uint32_t *code, uint32_t dim, const float *rq) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) {
        float bit_val = ((code[i / 32] >> (i % 32)) & 1u) ? 1.0f : -1.0f;
        sum += bit_val * rq[i];
    }
    return sum;
}

/* estimate_distance_fast, rabitq.rs:335-363. */
float orc_rabitq_estimate_fast(const uint32_t *code, float norm, float ipoo, uint32_t dim, float qnorm,
                               const float *rq) {
    if (norm < 1e-10f || qnorm < 1e-10f) {
        if (norm < 1e-10f && qnorm < 1e-10f) return 0.0f;
        if (norm < 1e-10f) return qnorm * qnorm;
        return norm * norm;
    }
    float ip = orc_rabitq_ip_fast(code, dim, rq);
    float est_ip = ip / ipoo;
    float dn = 2.0f - 2.0f * est_ip;
    return dn * norm * qnorm;
}

/* Query half of estimate_distances_batch_fast, rabitq.rs:398-420. Returns qnorm; rq is only
 * written when qnorm >= 1e-10. */
float orc_rabitq_query_prep(const float *rot_t, uint32_t dim, const float *q, float *rq) {
    float qnorm = orc_norm_seq(q, dim);
    if (qnorm < 1e-10f) return qnorm;
    float *qn = (float *)malloc(sizeof(float) * dim);
    for (uint32_t j = 0; j < dim; ++j) qn[j] = q[j] / qnorm;
    orc_rabitq_rotate(rot_t, dim, qn, rq);
    free(qn);
    return qnorm;
}

/* estimate_distances_batch_fast, rabitq.rs:398-436, over positions pos[0..n). */
void orc_rabitq_estimate_positions(const cuddle_index_desc *d, const float *q, const uint64_t *pos,
                                   uint64_t n, float *out) {
    uint32_t dim = d->dim, words = (dim + 31) / 32;
    float *rq = (float *)malloc(sizeof(float) * dim);
    float qnorm = orc_rabitq_query_prep(d->rabitq_rot_t, dim, q, rq);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = pos[i];
        if (qnorm < 1e-10f) { /* :408-413 */
            out[i] = d->rabitq_norm[p] * d->rabitq_norm[p];
        } else {
            out[i] = orc_rabitq_estimate_fast(d->rabitq_codes + p * words, d->rabitq_norm[p],
                                              d->rabitq_ip_oo[p], dim, qnorm, rq);
        }
    }
    free(rq);
}

/* ---------------------------------------------------------------------------------------- */
/* src/index/hierarchical.rs, src/index/hierarchical_with_rabitq.rs — search                  */
/* ---------------------------------------------------------------------------------------- */

/* Optional per-stage outputs of one query (any pointer may be NULL; *_cap in elements). */
typedef struct orc_trace {
    uint32_t *leaves;          /* accumulated leaf node ids, accumulation order */
    uint32_t leaves_cap, n_leaves;
    uint32_t *frontier_sizes;  /* SearchStats::probes_per_level (hierarchical.rs:861) */
    uint32_t levels_cap, n_levels;
    uint64_t *cand_pos;        /* all candidates in scan order: leaf-contiguous position */
    uint32_t *cand_key;        /* Hamming distance, or f32 bits of the RaBitQ estimate */
    uint64_t cand_cap, n_cand;
    uint64_t *sel_pos;         /* phase-1 selection, phase-1 order */
    uint32_t *sel_key;
    float *sel_dist;           /* reranked distance of each selected candidate */
    uint64_t sel_cap, n_sel;
} orc_trace;

typedef struct { float d; uint32_t node; uint32_t pos; } orc_nd;
static int orc_nd_cmp(const void *pa, const void *pb) {
    const orc_nd *a = (const orc_nd *)pa, *b = (const orc_nd *)pb;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->pos > b->pos) - (a->pos < b->pos);
}

typedef struct { uint32_t key; float fkey; float dist; uint64_t pos; uint64_t order; } orc_cand;
static int orc_cand_cmp_u32(const void *pa, const void *pb) {
    const orc_cand *a = (const orc_cand *)pa, *b = (const orc_cand *)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return (a->order > b->order) - (a->order < b->order);
}
static int orc_cand_cmp_f32(const void *pa, const void *pb) {
    const orc_cand *a = (const orc_cand *)pa, *b = (const orc_cand *)pb;
    if (a->fkey < b->fkey) return -1;
    if (a->fkey > b->fkey) return 1;
    return (a->order > b->order) - (a->order < b->order);
}
static int orc_cand_cmp_dist(const void *pa, const void *pb) {
    const orc_cand *a = (const orc_cand *)pa, *b = (const orc_cand *)pb;
    if (a->dist < b->dist) return -1;
    if (a->dist > b->dist) return 1;
    return (a->order > b->order) - (a->order < b->order);
}

static int orc_is_leaf(const cuddle_index_desc *d, uint32_t node) {
    return d->child_offsets[node + 1] == d->child_offsets[node]; /* hierarchical.rs:413 */
}

/* Routing loop of search (hierarchical.rs:349-423; twin hierarchical_with_rabitq.rs:345-419)
 * or of search_with_stats (:860-909) when CUDDLE_SEARCH_STATS_VARIANT is set.
 * leaves: capacity n_nodes. */
static int orc_route_impl(const cuddle_index_desc *d, const float *q, uint32_t probes, uint32_t flags,
                          uint32_t *leaves, uint32_t *n_leaves_out, orc_trace *tr) {
    const int stats = (flags & CUDDLE_SEARCH_STATS_VARIANT) != 0;
    const uint32_t dim = d->dim;
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * (d->n_nodes + 1));
    uint32_t *nxt = (uint32_t *)malloc(sizeof(uint32_t) * (d->n_nodes + 1));
    orc_nd *nd = (orc_nd *)malloc(sizeof(orc_nd) * (d->n_nodes + 1));
    const float **rows = (const float **)malloc(sizeof(float *) * 128);
    float *bd = (float *)malloc(sizeof(float) * 128);
    if (!cur || !nxt || !nd || !rows || !bd) return ORC_ERR_OOM;
    uint32_t ncur = d->n_roots, nl = 0;
    int rc = ORC_OK;
    memcpy(cur, d->roots, sizeof(uint32_t) * ncur);
    if (tr) tr->n_levels = 0;
    for (;;) {
        if (tr && tr->frontier_sizes && tr->n_levels < tr->levels_cap) tr->frontier_sizes[tr->n_levels++] = ncur;
        if (ncur == 0) break;
        /* distances: hierarchical.rs:355-391 (search) or :868-886 (stats) */
        if (!stats && ncur >= 4 && ncur <= 100) {
            for (uint32_t i = 0; i < ncur; ++i) rows[i] = d->centroids + (uint64_t)cur[i] * dim;
            orc_l2_batch(q, rows, ncur, dim, bd);
            for (uint32_t i = 0; i < ncur; ++i) nd[i].d = bd[i];
        } else {
            /* the reference splits this loop over its thread pool when the frontier exceeds 100 nodes (:355-365);
             * a no-op inside the one-query-per-thread batch driver (nested regions stay serial) */
#pragma omp parallel for schedule(static) if (ncur > 100)
            for (int64_t i = 0; i < (int64_t)ncur; ++i)
                nd[i].d = orc_distance(q, d->centroids + (uint64_t)cur[i] * dim, dim, (int)d->metric);
        }
        for (uint32_t i = 0; i < ncur; ++i) { nd[i].node = cur[i]; nd[i].pos = i; }
        uint32_t nsel;
        if (!stats && ncur <= probes) {
            nsel = ncur; /* :395-397 keep all, frontier order */
        } else {
            /* :399-407 (canonical order = the stable sort of :888-894) */
            if (ncur >= 2)
                for (uint32_t i = 0; i < ncur; ++i)
                    if (nd[i].d != nd[i].d) { rc = ORC_ERR_NAN; goto done; }
            qsort(nd, ncur, sizeof(orc_nd), orc_nd_cmp);
            nsel = ncur < probes ? ncur : probes;
        }
        /* partition + expand, :411-422 */
        uint32_t nn = 0;
        for (uint32_t i = 0; i < nsel; ++i) {
            uint32_t node = nd[i].node;
            if (orc_is_leaf(d, node)) leaves[nl++] = node;
        }
        for (uint32_t i = 0; i < nsel; ++i) {
            uint32_t node = nd[i].node;
            if (!orc_is_leaf(d, node))
                for (uint32_t c = d->child_offsets[node]; c < d->child_offsets[node + 1]; ++c) nxt[nn++] = d->children[c];
        }
        uint32_t *t = cur; cur = nxt; nxt = t;
        ncur = nn;
    }
done:
    *n_leaves_out = nl;
    free(cur); free(nxt); free(nd); free((void *)rows); free(bd);
    return rc;
}

/* search_leaves (hierarchical.rs:439-464 / hierarchical_with_rabitq.rs:435-459) with the
 * canonical tie rule; stats variant per :925-935. */
static int orc_search_leaves_impl(const cuddle_index_desc *d, const float *q, const uint32_t *leaves,
                                  uint32_t n_leaves, uint32_t k, uint32_t rerank_factor, uint32_t flags,
                                  uint64_t *out_ids, float *out_dist, uint32_t *out_count, orc_trace *tr) {
    const int stats = (flags & CUDDLE_SEARCH_STATS_VARIANT) != 0;
    const uint32_t dim = d->dim;
    *out_count = 0;
    if (tr) { tr->n_cand = 0; tr->n_sel = 0; }
    uint64_t ncand = 0;
    for (uint32_t l = 0; l < n_leaves; ++l) ncand += d->vec_offsets[leaves[l] + 1] - d->vec_offsets[leaves[l]];
    if (ncand == 0) return ORC_OK; /* :452-454 */
    uint64_t rerank_k = (uint64_t)k * rerank_factor;
    if (!stats && rerank_k > 10000) rerank_k = 10000; /* :447 vs :927 */
    orc_cand *c = (orc_cand *)malloc(sizeof(orc_cand) * ncand);
    if (!c) return ORC_ERR_OOM;
    int rc = ORC_OK;
    /* phase 1: collect_leaf_candidates, hierarchical.rs:618-641 / rabitq :605-619 */
    uint64_t n = 0;
    for (uint32_t l = 0; l < n_leaves; ++l)
        for (uint64_t p = d->vec_offsets[leaves[l]]; p < d->vec_offsets[leaves[l] + 1]; ++p) {
            c[n].pos = p; c[n].order = n; c[n].key = 0; c[n].fkey = 0.0f; c[n].dist = 0.0f;
            ++n;
        }
    if (d->code_kind == CUDDLE_CODE_BINARY) {
        uint32_t words = (dim + 63) / 64;
        uint64_t *qc = (uint64_t *)malloc(sizeof(uint64_t) * words);
        orc_binary_quantize(q, 1, dim, d->threshold, qc); /* hierarchical.rs:340 */
        /* per-leaf tasks, and per-vector ones in leaves of more than 100 vectors, in the reference (:591-641) */
#pragma omp parallel for schedule(static) if (n > 100)
        for (int64_t i = 0; i < (int64_t)n; ++i) c[i].key = orc_hamming(qc, d->codes + c[i].pos * words, words);
        free(qc);
    } else {
        uint64_t *pos = (uint64_t *)malloc(sizeof(uint64_t) * n);
        float *est = (float *)malloc(sizeof(float) * n);
        for (uint64_t i = 0; i < n; ++i) pos[i] = c[i].pos;
        orc_rabitq_estimate_positions(d, q, pos, n, est);
        for (uint64_t i = 0; i < n; ++i) {
            c[i].fkey = est[i];
            memcpy(&c[i].key, &est[i], 4);
            if (est[i] != est[i]) rc = ORC_ERR_NAN;
        }
        free(pos); free(est);
        if (rc) goto done;
    }
    if (tr && tr->cand_pos && tr->cand_key) {
        tr->n_cand = n;
        for (uint64_t i = 0; i < n && i < tr->cand_cap; ++i) { tr->cand_pos[i] = c[i].pos; tr->cand_key[i] = c[i].key; }
    } else if (tr) {
        tr->n_cand = n;
    }
    /* select_top_candidates, hierarchical.rs:645-655 / rabitq :623-633; canonical = stable sort + truncate */
    qsort(c, n, sizeof(orc_cand), d->code_kind == CUDDLE_CODE_BINARY ? orc_cand_cmp_u32 : orc_cand_cmp_f32);
    uint64_t nsel = n < rerank_k ? n : rerank_k;
    /* rerank_with_full_precision, hierarchical.rs:659-674 */
    int nan_seen = 0; /* the reference reranks with par_iter (:659-674) */
#pragma omp parallel for schedule(static) reduction(| : nan_seen) if (nsel > 100)
    for (int64_t i = 0; i < (int64_t)nsel; ++i) {
        c[i].order = (uint64_t)i; /* phase-1 order */
        c[i].dist = orc_distance(q, d->vectors + c[i].pos * dim, dim, (int)d->metric);
        if (c[i].dist != c[i].dist) nan_seen = 1;
    }
    if (nan_seen) rc = ORC_ERR_NAN;
    if (tr) {
        tr->n_sel = nsel;
        for (uint64_t i = 0; i < nsel && i < tr->sel_cap; ++i) {
            if (tr->sel_pos) tr->sel_pos[i] = c[i].pos;
            if (tr->sel_key) tr->sel_key[i] = c[i].key;
            if (tr->sel_dist) tr->sel_dist[i] = c[i].dist;
        }
    }
    if (rc) goto done;
    /* finalize_results, hierarchical.rs:678-691 */
    qsort(c, nsel, sizeof(orc_cand), orc_cand_cmp_dist);
    uint32_t nout = nsel < k ? (uint32_t)nsel : k;
    for (uint32_t i = 0; i < nout; ++i) { out_ids[i] = d->ids[c[i].pos]; out_dist[i] = c[i].dist; }
    *out_count = nout;
done:
    free(c);
    return rc;
}

static void orc_fill_unused(uint64_t *out_ids, float *out_dist, uint32_t from, uint32_t k) {
    for (uint32_t i = from; i < k; ++i) { out_ids[i] = UINT64_MAX; out_dist[i] = INFINITY; }
}

/* ClusteredIndex::search / ClusteredIndexWithRaBitQ::search for one query. out_ids/out_dist: [k]. */
int orc_search(const cuddle_index_desc *d, const float *q, uint32_t dim, uint32_t k, uint32_t probes,
               uint32_t rerank_factor, uint32_t flags, uint64_t *out_ids, float *out_dist,
               uint32_t *out_count, orc_trace *tr) {
    if (!d || !q || !out_ids || !out_dist || !out_count) return ORC_ERR_BAD_ARG;
    if (dim != d->dim) return ORC_ERR_BAD_DIM; /* hierarchical.rs:337 */
    if (k == 0 || probes == 0 || rerank_factor == 0) return ORC_ERR_BAD_ARG; /* k-1 underflow, :651 */
    uint32_t *leaves = (uint32_t *)malloc(sizeof(uint32_t) * (d->n_nodes + 1));
    if (!leaves) return ORC_ERR_OOM;
    uint32_t nl = 0;
    *out_count = 0;
    int rc = orc_route_impl(d, q, probes, flags, leaves, &nl, tr);
    if (tr) {
        tr->n_leaves = nl;
        if (tr->leaves) for (uint32_t i = 0; i < nl && i < tr->leaves_cap; ++i) tr->leaves[i] = leaves[i];
    }
    if (rc == ORC_OK && nl > 0) /* :426-430 */
        rc = orc_search_leaves_impl(d, q, leaves, nl, k, rerank_factor, flags, out_ids, out_dist, out_count, tr);
    orc_fill_unused(out_ids, out_dist, *out_count, k);
    free(leaves);
    return rc;
}

/* search_leaves for a caller-given leaf list (stage parity). */
int orc_search_leaves(const cuddle_index_desc *d, const float *q, const uint32_t *leaves, uint32_t n_leaves,
                      uint32_t k, uint32_t rerank_factor, uint32_t flags, uint64_t *out_ids, float *out_dist,
                      uint32_t *out_count, orc_trace *tr) {
    if (k == 0 || rerank_factor == 0) return ORC_ERR_BAD_ARG;
    int rc = orc_search_leaves_impl(d, q, leaves, n_leaves, k, rerank_factor, flags, out_ids, out_dist, out_count, tr);
    orc_fill_unused(out_ids, out_dist, *out_count, k);
    return rc;
}

/* Batch driver: one query per thread (the CPU baseline of the benchmark; the reference has no
 * batch API — benches/profile_bench.rs:286-296 — so this is the "one search() per query over
 * a thread pool" shape). n_threads <= 0 uses the OpenMP default. Returns the first error. */
int orc_search_batch(const cuddle_index_desc *d, const float *queries, uint32_t n_queries, uint32_t dim,
                     uint32_t k, uint32_t probes, uint32_t rerank_factor, uint32_t flags, uint64_t *out_ids,
                     float *out_dist, uint32_t *out_count, int n_threads) {
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t qi = 0; qi < (int64_t)n_queries; ++qi) {
        uint32_t cnt = 0;
        int rc = orc_search(d, queries + (uint64_t)qi * dim, dim, k, probes, rerank_factor, flags,
                            out_ids + (uint64_t)qi * k, out_dist + (uint64_t)qi * k, &cnt, NULL);
        if (out_count) out_count[qi] = cnt;
        if (rc != ORC_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            { if (rc_all == ORC_OK) rc_all = rc; }
        }
    }
    return rc_all;
}

/* The reference's shape: search() calls issued one after another, each spreading its routing, scan
 * and rerank loops over the thread pool (the parallel regions inside orc_search). */
int orc_search_sequential(const cuddle_index_desc *d, const float *queries, uint32_t n_queries, uint32_t dim,
                          uint32_t k, uint32_t probes, uint32_t rerank_factor, uint32_t flags, uint64_t *out_ids,
                          float *out_dist, uint32_t *out_count, int n_threads) {
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
        uint32_t cnt = 0;
        int rc = orc_search(d, queries + (uint64_t)qi * dim, dim, k, probes, rerank_factor, flags,
                            out_ids + (uint64_t)qi * k, out_dist + (uint64_t)qi * k, &cnt, NULL);
        if (out_count) out_count[qi] = cnt;
        if (rc != ORC_OK && rc_all == ORC_OK) rc_all = rc;
    }
    return rc_all;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
