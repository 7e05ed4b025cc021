/*
 * cuddle_hip.h — C ABI of libcuddle_hip.so: the MI355X (gfx950) implementation of
 * CuddleDB's batched two-phase ANN query path
 *   centroid routing -> binary-code (Hamming) / RaBitQ leaf scan -> fp32 rerank -> top-k.
 *
 * The reference (LLay/vectordb, Rust) has no FFI layer for this path; the boundary it
 * exposes is the inherent method
 *   ClusteredIndex::search(&self, query:&[f32], k, probes_per_level, rerank_factor)
 *       -> Vec<(usize, f32)>                       src/index/hierarchical.rs:330-336
 *   ClusteredIndexWithRaBitQ::search(...)          src/index/hierarchical_with_rabitq.rs:329-335
 * Every entry point below names the reference function(s) it replaces. A Rust caller
 * binds these with `extern "C"` declarations and links libcuddle_hip.so from its build.rs.
 *
 * Conventions
 *   - plain pointers and sizes only; all arrays are row-major, little-endian;
 *   - "host" entry points take host pointers and synchronise before returning;
 *     "_device" entry points take device pointers + a hipStream_t (as void*) and are
 *     asynchronous on that stream;
 *   - every function returns a cuddle_status; cuddle_last_error() gives the message of the
 *     last failure on the calling thread; the library never aborts the process
 *     (the reference panics/aborts: hierarchical.rs:337,402; Cargo.toml:54);
 *   - a handle may be used from several threads: an internal mutex serialises the calls while
 *     they ENQUEUE. A handle has one work space, so batches of one handle never overlap on the
 *     GPU either: a call that names another stream than the handle's previous call first waits,
 *     on the host, for that previous stream (use one stream per handle to avoid the wait, and
 *     several handles - e.g. shards - for concurrent batches);
 *   - limits (CUDDLE_ERR_UNSUPPORTED beyond them): dim <= 32768; fewer than 2^32 vectors per
 *     handle; at most 262144 vectors in one leaf; fewer than 2^25 candidates per query
 *     (leaf-list capacity x largest leaf); probes_per_level <= 4096 when a level is wider than
 *     4096 nodes; in the (uncapped) stats variant min(k * rerank_factor, the candidates a query can reach) <= 16384.
 */
#ifndef CUDDLE_HIP_H
#define CUDDLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUDDLE_ABI_VERSION 2

typedef enum cuddle_status {
    CUDDLE_OK = 0,
    CUDDLE_ERR_BAD_ARG = 1,   /* null pointer, k == 0, probes == 0, sizes out of supported range */
    CUDDLE_ERR_BAD_DIM = 2,   /* query dimension != index dimension (reference: assert, hierarchical.rs:337) */
    CUDDLE_ERR_HIP = 3,       /* a HIP runtime call failed */
    CUDDLE_ERR_NO_DEVICE = 4, /* no gfx950 device visible */
    CUDDLE_ERR_OOM = 5,       /* device or host allocation failed */
    CUDDLE_ERR_BAD_INDEX = 6, /* index description is inconsistent (offsets not monotone, ids out of range ...) */
    CUDDLE_ERR_UNSUPPORTED = 7, /* valid in the reference but outside this build's limits (documented per call) */
    CUDDLE_ERR_RCCL = 8        /* RCCL could not be loaded, or a collective failed */
} cuddle_status;

/* DistanceMetric, src/distance/mod.rs:10-17. Returned distances follow distance() (:23-34):
 * L2 -> sum (a-b)^2 (squared, no sqrt); DOT -> -sum a*b; COSINE -> 1 - dot/(|a||b|). */
typedef enum cuddle_metric { CUDDLE_METRIC_L2 = 0, CUDDLE_METRIC_COSINE = 1, CUDDLE_METRIC_DOT = 2 } cuddle_metric;

/* Phase-1 code family: BinaryQuantizer/BinaryVector (src/quantization/binary.rs:16-21,101-106)
 * or RaBitQQuantizer/RaBitQVector (src/quantization/rabitq.rs:16-32). */
typedef enum cuddle_code_kind { CUDDLE_CODE_BINARY = 0, CUDDLE_CODE_RABITQ = 1 } cuddle_code_kind;

/* Search behaviour flags. Default (0) follows `search` (hierarchical.rs:330-431):
 *   - levels with <= probes nodes keep all nodes in frontier order (:395-397);
 *   - rerank_k = min(k*rerank_factor, 10000) (:447);
 *   - levels with 4..=100 nodes are routed with L2^2 whatever the metric (:366-380).
 * CUDDLE_SEARCH_STATS_VARIANT follows `search_with_stats` (:831-938) instead: every level is
 * stably sorted, no 10000 cap (:927), routing always uses the index metric (:868-886). */
#define CUDDLE_SEARCH_STATS_VARIANT 0x1u
/* Record per-stage HIP events on the call's stream; read them with cuddle_stage_times(). */
#define CUDDLE_SEARCH_TIME_STAGES 0x2u
/* Route dense levels with the MFMA contraction + exact refinement (default) or force the
 * all-exact VALU routing kernel (bit-identical results; used for A/B measurements). */
#define CUDDLE_SEARCH_ROUTE_EXACT_ONLY 0x4u
/* Evaluate every level whose frontiers differ per query grouped by parent node, even where the
 * library would judge the frontiers too narrow for that to pay (bit-identical results; tests). */
#define CUDDLE_SEARCH_ROUTE_GROUPED 0x8u
/* A level that every query sees whole is filtered with single-pass half-precision matrix-core
 * products where the index holds operands for it (levels below a level of at most 4096 nodes);
 * this flag keeps the three-product split-bf16 filter there (bit-identical results; A/B, tests). */
#define CUDDLE_SEARCH_ROUTE_SPLIT_FILTER 0x10u

/* Flat description of one index (or one leaf shard of it). Mirrors ClusteredIndex's public
 * fields nodes/root_ids/quantizer/binary_vectors/full_vectors (hierarchical.rs:51-70) and
 * ClusterNode (:18-32), flattened:
 *   node i: centroid = centroids[i*dim .. ), children = children[child_offsets[i] .. child_offsets[i+1]),
 *           vector range (leaf-contiguous positions) = [vec_offsets[i], vec_offsets[i+1]).
 *   A node without children is a leaf (hierarchical.rs:413). Internal nodes have an empty
 *   vector range. Position p (0 <= p < n_vectors) is the p-th vector in "leaf-contiguous
 *   order": leaves in node-id order, each leaf's vector_indices in stored order;
 *   ids[p] is the reference's vector index (usize) of that vector.
 * All per-vector arrays (ids, vectors, codes, rabitq_*) are given in leaf-contiguous order.
 * The library copies everything to the device at create time; the caller keeps ownership. */
/* cuddle_index_desc.flags */
/* centroids, ids, vectors, codes and the rabitq_* arrays are device pointers on the target device
 * (the tree arrays child_offsets / children / roots / vec_offsets are always host pointers). */
#define CUDDLE_DESC_DEVICE_ARRAYS 0x1u
/* with DEVICE_ARRAYS: use `vectors` and `ids` in place instead of copying them; the caller keeps
 * them alive and unchanged until cuddle_index_destroy (a 64 GB shard is then held once, not twice). */
#define CUDDLE_DESC_ADOPT_VECTORS 0x2u

typedef struct cuddle_index_desc {
    uint32_t abi_version;  /* CUDDLE_ABI_VERSION */
    uint32_t dim;          /* D */
    uint32_t metric;       /* cuddle_metric */
    uint32_t code_kind;    /* cuddle_code_kind */
    uint32_t n_nodes;
    uint32_t n_roots;
    uint64_t n_vectors;    /* vectors stored in THIS shard (sum of leaf sizes) */
    const float *centroids;         /* [n_nodes][dim]  ClusterNode::full_centroid */
    const uint32_t *child_offsets;  /* [n_nodes+1] */
    const uint32_t *children;       /* [child_offsets[n_nodes]] node ids */
    const uint32_t *roots;          /* [n_roots]  ClusteredIndex::root_ids */
    const uint64_t *vec_offsets;    /* [n_nodes+1] */
    const uint64_t *ids;            /* [n_vectors] */
    const float *vectors;           /* [n_vectors][dim]  MmapVectorStore rows (src/storage/mmap.rs:106-115) */
    /* binary codes: BinaryVector::bits, ceil(dim/64) u64 words per vector, bit i of the
     * vector at word i/64, bit i%64 (binary.rs:150-155); threshold = BinaryQuantizer::threshold */
    const uint64_t *codes;          /* [n_vectors][ceil(dim/64)] or NULL for RABITQ */
    float threshold;
    uint32_t flags;                 /* CUDDLE_DESC_* */
    /* RaBitQ: rotation_matrix_t rows (rabitq.rs:20), RaBitQVector fields (:24-32) */
    const float *rabitq_rot_t;      /* [dim][dim] or NULL */
    const uint32_t *rabitq_codes;   /* [n_vectors][ceil(dim/32)] */
    const float *rabitq_norm;       /* [n_vectors] */
    const float *rabitq_ip_oo;      /* [n_vectors] */
} cuddle_index_desc;

typedef struct cuddle_index cuddle_index; /* opaque device-resident index + workspace */

/* SearchStats (hierarchical.rs:36-45), summed over the batch plus per-query maxima. */
typedef struct cuddle_stats {
    uint64_t leaves_searched;        /* sum over queries */
    uint64_t vectors_scanned;        /* sum over queries of candidates (vectors_scanned_binary) */
    uint64_t vectors_reranked;       /* sum over queries (vectors_reranked_full) */
    uint32_t max_candidates;         /* max over queries */
    uint32_t max_leaves;             /* max over queries */
    uint32_t tree_levels;            /* routing iterations executed */
    uint32_t route_fallbacks;        /* queries x levels re-routed by the all-exact kernel */
    /* RaBitQ phase 1: candidates whose estimate was evaluated in the reference's sequential arithmetic
     * (the others were excluded by a proven bound), and queries that evaluated all of theirs */
    uint64_t rabitq_exact_estimates;
    uint64_t rabitq_exact_all_queries;
} cuddle_stats;

/* Stage indices for cuddle_stage_times(). */
enum {
    CUDDLE_STAGE_QUANTIZE = 0, /* R1   query binarisation / RaBitQ query prep */
    CUDDLE_STAGE_ROUTE = 1,    /* R2+R3 all tree levels */
    CUDDLE_STAGE_SCAN = 2,     /* R4+R5 the leaf scan + phase-1 selection kernel(s) */
    CUDDLE_STAGE_RERANK = 3,   /* R6 */
    CUDDLE_STAGE_TOPK = 4,     /* R7 */
    CUDDLE_STAGE_SCAN_PREP = 5, /* candidate counts per query and the order in which the scan takes the queries */
    CUDDLE_STAGE_COUNT = 6
};

/* ---- library ---------------------------------------------------------------------------- */
uint32_t cuddle_abi_version(void);
const char *cuddle_last_error(void);
/* Number of visible HIP devices (0 when none); never fails. */
int cuddle_device_count(void);

/* ---- index life cycle ------------------------------------------------------------------- */
/* Replaces ClusteredIndex::build's in-memory result as the thing `search` runs on
 * (hierarchical.rs:163-173): uploads the flat arrays to device `device_id`, re-lays the codes
 * out per leaf for coalesced streaming, precomputes centroid norms. */
cuddle_status cuddle_index_create(const cuddle_index_desc *desc, int device_id, cuddle_index **out);
void cuddle_index_destroy(cuddle_index *idx);
/* Pre-size the handle's device workspace so that later searches with batch <= max_queries,
 * k <= max_k, probes <= max_probes, k*rerank_factor <= max_rerank_k allocate nothing. */
cuddle_status cuddle_index_reserve(cuddle_index *idx, uint32_t max_queries, uint32_t max_k,
                                   uint32_t max_probes, uint32_t max_rerank_k);

/* ---- the hot path ----------------------------------------------------------------------- */
/* Batched `search` (hierarchical.rs:330-431 / hierarchical_with_rabitq.rs:329-427) for
 * n_queries independent queries; row q of the outputs is what search(&queries[q], k, probes,
 * rerank_factor) returns: ascending distance, out_count[q] <= k results (fewer when the
 * reached leaves hold fewer vectors, hierarchical.rs:426-428,452-454). Unused slots hold
 * id = UINT64_MAX, dist = +inf. Ties: phase-1 by (code distance, scan position), final by
 * (distance, phase-1 order) — the stable-sort order of search_with_stats (:930-932).
 * queries: [n_queries][dim] host; out_ids/out_dist: [n_queries][k] host; out_count: [n_queries]
 * host or NULL; stats may be NULL. */
cuddle_status cuddle_search_batch(cuddle_index *idx, const float *queries, uint32_t n_queries,
                                  uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                  uint32_t rerank_factor, uint32_t flags, uint64_t *out_ids,
                                  float *out_dist, uint32_t *out_count, cuddle_stats *stats);

/* Same, with every pointer a device pointer on the index's device and the work enqueued on
 * `stream` (a hipStream_t; NULL = the null stream). Returns after enqueueing; results are
 * valid once the stream has been synchronised. d_out_count may be NULL. */
cuddle_status cuddle_search_batch_device(cuddle_index *idx, const float *d_queries,
                                         uint32_t n_queries, uint32_t dim, uint32_t k,
                                         uint32_t probes_per_level, uint32_t rerank_factor,
                                         uint32_t flags, uint64_t *d_out_ids, float *d_out_dist,
                                         uint32_t *d_out_count, void *stream);

/* The same search for callers that must not allocate device memory while they run (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc). It has no counterpart in
 * the reference. Where cuddle_search_batch_device grows the handle's work space on demand, this
 * call lays the work space out in a fixed 32 MiB array that every GPU the library runs on holds
 * from the library's load until the process ends; it calls no hipMalloc / hipFree and records no
 * stage timings (CUDDLE_SEARCH_TIME_STAGES is ignored). Batches are cut into chunks that fit the
 * array; CUDDLE_ERR_UNSUPPORTED when the work space of a single query does not fit (very wide
 * leaf lists). Results are bit-identical to cuddle_search_batch_device. The array is shared by
 * every handle on the GPU and keeps the previous call's contents: calls that overlap on one GPU
 * (two streams, two handles, two threads) race; a call on another stream than the array's
 * previous user waits for that stream on the host first. */
cuddle_status cuddle_search_batch_device_noalloc(cuddle_index *idx, const float *d_queries,
                                                 uint32_t n_queries, uint32_t dim, uint32_t k,
                                                 uint32_t probes_per_level, uint32_t rerank_factor,
                                                 uint32_t flags, uint64_t *d_out_ids,
                                                 float *d_out_dist, uint32_t *d_out_count,
                                                 void *stream);

/* Batched search_with_stats (hierarchical.rs:831-938): the always-sorting variant of the search
 * (every level stably sorted, routing always by the index metric, no 10000 cap) together with
 * what SearchStats (:36-45) records for each query. Host pointers; row q of every output
 * belongs to query q:
 *   out_ids / out_dist [n][k], out_count [n]        the results, as cuddle_search_batch
 *   out_leaves [n][leaf_stride], out_leaf_count [n]  leaves_searched_ids in accumulation order
 *                                                    (:902, :917) and leaves_searched (:916);
 *                                                    leaf_stride >= cuddle_index_leaf_stride()
 *   out_level_nodes [n][level_stride], out_level_count [n]
 *                                                    probes_per_level (:861): the frontier size at
 *                                                    the top of every loop iteration, including the
 *                                                    empty frontier that ends the loop;
 *                                                    level_stride >= cuddle_index_levels() + 1
 *   out_scanned [n]                                  vectors_scanned_binary (:920-922)
 *   out_reranked [n]                                 vectors_reranked_full (:927-928)
 * total_leaves, total_vectors and tree_depth are properties of the index (cuddle_index_levels()
 * - 1 is the depth of the deepest node). */
cuddle_status cuddle_search_batch_with_stats(cuddle_index *idx, const float *queries,
                                             uint32_t n_queries, uint32_t dim, uint32_t k,
                                             uint32_t probes, uint32_t rerank_factor,
                                             uint64_t *out_ids, float *out_dist, uint32_t *out_count,
                                             uint32_t *out_leaves, uint32_t leaf_stride,
                                             uint32_t *out_leaf_count, uint32_t *out_level_nodes,
                                             uint32_t level_stride, uint32_t *out_level_count,
                                             uint32_t *out_scanned, uint32_t *out_reranked);

/* Stats of the most recent search on this handle (synchronises the handle's last stream). */
cuddle_status cuddle_last_stats(cuddle_index *idx, cuddle_stats *stats);
/* Per-stage device time in milliseconds, summed over every search on this handle that ran with
 * CUDDLE_SEARCH_TIME_STAGES since the previous call of this function; ms has CUDDLE_STAGE_COUNT
 * entries, *n_searches (may be NULL) receives how many searches the sums cover. Waits for the
 * last of them to finish. */
cuddle_status cuddle_stage_times(cuddle_index *idx, float *ms, uint32_t *n_searches);

/* ---- stage entry points (host pointers; used by the parity tests) ----------------------- */
/* distance() for n independent pairs (src/distance/mod.rs:23-34; accumulation order of
 * simd_neon.rs:13-140): out[i] = distance(a[i], b[i], metric); a, b: [n][dim]. */
cuddle_status cuddle_distance_pairs(const float *a, const float *b, uint64_t n, uint32_t dim,
                                    uint32_t metric, float *out, int device_id);
/* BinaryQuantizer::quantize (binary.rs:144-159) for n vectors: codes [n][ceil(dim/64)]. */
cuddle_status cuddle_binary_quantize(const float *vectors, uint64_t n, uint32_t dim,
                                     float threshold, uint64_t *codes, int device_id);
/* hamming_distance (binary.rs:184-186) for n pairs of codes of `words` u64 each. */
cuddle_status cuddle_hamming_pairs(const uint64_t *a, const uint64_t *b, uint64_t n,
                                   uint32_t words, uint32_t *out, int device_id);
/* BinaryQuantizer::from_vectors threshold (binary.rs:122-141): sequential f32 sum / count.
 * Host computation (index-build side input of the path). */
float cuddle_binary_threshold(const float *vectors, uint64_t n, uint32_t dim);

/* Routing only (hierarchical.rs:349-423): for each query the accumulated leaf node ids in
 * accumulation order. out_leaves: [n_queries][leaf_stride]; out_leaf_count: [n_queries].
 * leaf_stride must be >= probes * tree_levels (cuddle_index_levels). */
cuddle_status cuddle_route_batch(cuddle_index *idx, const float *queries, uint32_t n_queries,
                                 uint32_t dim, uint32_t probes_per_level, uint32_t flags,
                                 uint32_t *out_leaves, uint32_t leaf_stride,
                                 uint32_t *out_leaf_count);
/* ---- index-build side --------------------------------------------------------------------- */
/* KMeans::assign (src/clustering/kmeans.rs:136-160; nearest_centroid, :229-236, is the one-vector
 * case): for each of n vectors the index of the nearest of k centroids under distance(metric)
 * (src/distance/mod.rs:23-34) and that distance; equal minima go to the lowest index, as
 * Iterator::min_by does. The n x k matrix is evaluated approximately on the matrix cores and
 * exactly only where the approximation's error bound leaves a choice, so the result is the
 * reference's bit for bit. All pointers are device memory: d_vectors [n][dim], d_centroids
 * [k][dim], d_assign [n], d_dist [n]. *out_fallbacks (host, may be NULL) receives the number of
 * (vector, 256-centroid segment) pairs that had to be compared exactly in full (three or more
 * near-tied candidates in one segment, e.g. duplicate centroids).
 * Allocates its workspace per call and returns after the work on `stream` has finished. */
cuddle_status cuddle_kmeans_assign(const float *d_vectors, uint64_t n, const float *d_centroids,
                                   uint32_t k, uint32_t dim, uint32_t metric, uint32_t *d_assign,
                                   float *d_dist, uint64_t *out_fallbacks, int device_id,
                                   void *stream);

/* KMeans::update_centroids (src/clustering/kmeans.rs:163-215): d_centroids[c] = the mean of the vectors
 * with d_assign[v] == c, d_counts[c] = how many there are. The sum is formed as the reference forms it:
 * per chunk of 2048 consecutive vectors one sequential f32 accumulation in vector order (:170-183), the
 * chunks' sums merged in chunk order (rayon's reduce, :187-199, keeps the order and leaves the grouping
 * open; this merges from the left), then divided by the count as f32 (:209-213). The row of a cluster
 * without vectors is left untouched: the reference re-seeds it with a randomly drawn vector (:217-223),
 * which is the caller's draw to make. Assignments >= k are ignored. Device pointers: d_vectors [n][dim],
 * d_assign [n], d_centroids [k][dim], d_counts [k]. dim <= 32768. Allocates its workspace per call and
 * returns after the work on `stream` has finished. */
cuddle_status cuddle_kmeans_update(const float *d_vectors, uint64_t n, uint32_t dim, const uint32_t *d_assign,
                                   uint32_t k, float *d_centroids, uint32_t *d_counts, int device_id,
                                   void *stream);

/* KMeans::init_plusplus (src/clustering/kmeans.rs:36-88) with the reference's random draws passed in:
 * `first` = the index vectors.choose() returned (:49), uniforms[j - 1] (host, k - 1 values in [0, 1)) =
 * rand::random::<f32>() of round j (:68). Every round evaluates distance(v, newest centroid) for all
 * vectors on the device and keeps the running minimum (:55-64); the sum of the squared distances and
 * the walk that picks the next centroid (:67-77) are sequential single-precision chains and run on the
 * host, over the weights copied back each round. d_centroids [k][dim] (device) receives the chosen
 * rows, chosen [k] (host, may be NULL) their indices. Returns when done. */
cuddle_status cuddle_kmeanspp_init(const float *d_vectors, uint64_t n, uint32_t dim, uint32_t k,
                                   uint32_t metric, uint64_t first, const float *uniforms,
                                   float *d_centroids, uint64_t *chosen, int device_id, void *stream);

/* BinaryQuantizer::quantize (src/quantization/binary.rs:144-159) for n device-resident vectors:
 * d_codes [n][ceil(dim/64)]. Enqueued on `stream`. (cuddle_binary_quantize is the host-pointer form.) */
cuddle_status cuddle_binary_encode_device(const float *d_vectors, uint64_t n, uint32_t dim, float threshold,
                                          uint64_t *d_codes, int device_id, void *stream);

/* RaBitQQuantizer::quantize / quantize_batch (src/quantization/rabitq.rs:114-179) for n device-resident
 * vectors: d_codes [n][ceil(dim/32)], d_norm [n], d_ip_oo [n], every sum in the reference's order.
 * d_rot_t [dim][dim] = rotation_matrix_t (:20) on the device. Returns after the work on `stream` has
 * finished. */
cuddle_status cuddle_rabitq_encode_device(const float *d_rot_t, uint32_t dim, const float *d_vectors,
                                          uint64_t n, uint32_t *d_codes, float *d_norm, float *d_ip_oo,
                                          int device_id, void *stream);

/* Diagnostic of the matrix-core routing filter (no counterpart in the reference, whose routing
 * at hierarchical.rs:355-391 evaluates every distance exactly; here a level as wide as a whole
 * frontier is first ranked by approximate distances, and the selection is accepted only where an
 * error bound proves it equal to the exact one). For every (query, node of tree level `level`):
 * err = |approximate distance - distance evaluated in f64|. *out_max_abs = the largest err,
 * *out_max_ratio = the largest err / bound(query); the proof needs the ratio to stay below 1.
 * CUDDLE_ERR_UNSUPPORTED when the level has no filter operands (fewer than 512 nodes) or can
 * never be a whole frontier with this probes_per_level. */
cuddle_status cuddle_route_filter_error(cuddle_index *idx, const float *queries,
                                        uint32_t n_queries, uint32_t dim,
                                        uint32_t probes_per_level, uint32_t level,
                                        float *out_max_ratio, float *out_max_abs);
/* How a batch of n_queries would scan its leaves (no counterpart in the reference, whose scan is per query,
 * hierarchical.rs:591-641): *out_mode = 0 per query (one block streams the codes of a query's leaves), 1 leaf by leaf
 * (binary codes of 512 bits and more, when the batch reaches a leaf several times over: a leaf's codes are read once
 * per 64 queries and the distances handed to the per-query selection). The results are the same either way. */
cuddle_status cuddle_scan_mode(cuddle_index *idx, uint32_t n_queries, uint32_t probes_per_level,
                               uint32_t *out_mode);
/* Measuring aid, no counterpart in the reference: what the leaf scan's memory accesses cost by themselves.
 * d_queries: [n_batches][n_queries][dim] on the device (1..64 batches of one chunk each). Every batch is routed
 * once; then `reps` rounds over the batches are timed (HIP events, null stream), first with a kernel that issues
 * exactly the reads of the Hamming scan - the same 16-byte pieces of the same leaves, as many in flight - and
 * nothing else, then with the scan kernel itself. Going round several batches keeps a batch's leaves as cold as
 * they are between a caller's successive searches. *out_read_ms / *out_scan_ms = mean duration of one launch of
 * either; *out_code_bytes = the code bytes one launch covers on average (candidates x 8 x ceil(dim / 64), what
 * collect_leaf_candidates reads, hierarchical.rs:591-641). Binary codes of 2, 4, 8, 12, 15, 16 or 24 words;
 * CUDDLE_ERR_UNSUPPORTED otherwise. Synchronous. */
cuddle_status cuddle_scan_read_ceiling(cuddle_index *idx, const float *d_queries, uint32_t n_batches,
                                       uint32_t n_queries, uint32_t dim, uint32_t k,
                                       uint32_t probes_per_level, uint32_t rerank_factor, uint32_t reps,
                                       float *out_read_ms, float *out_scan_ms, uint64_t *out_code_bytes);
/* The same on device pointers, enqueued on `stream` (1..4096 queries per call): d_leaves
 * [n_queries][leaf_stride] and d_leaf_count [n_queries] receive the leaf lists. leaf_stride
 * must be at least cuddle_index_leaf_stride(probes_per_level), the capacity every search with
 * that probes_per_level uses for its leaf lists (it depends on the tree only). */
cuddle_status cuddle_index_leaf_stride(const cuddle_index *idx, uint32_t probes_per_level,
                                       uint32_t *out);
cuddle_status cuddle_route_batch_device(cuddle_index *idx, const float *d_queries,
                                        uint32_t n_queries, uint32_t dim,
                                        uint32_t probes_per_level, uint32_t flags,
                                        uint32_t *d_leaves, uint32_t leaf_stride,
                                        uint32_t *d_leaf_count, void *stream);
/* collect_leaf_candidates (hierarchical.rs:618-641 / hierarchical_with_rabitq.rs:605-619)
 * for given leaf lists: per query the (position, code distance) pairs in scan order.
 * out_pos: [n_queries][cand_stride] leaf-contiguous positions (ids[pos] is the vector id);
 * out_key: [n_queries][cand_stride] Hamming distance (BINARY, as u32) or the bit pattern of
 * the estimated distance (RABITQ, reinterpret as float). */
cuddle_status cuddle_scan_batch(cuddle_index *idx, const float *queries, uint32_t n_queries,
                                uint32_t dim, const uint32_t *leaves, uint32_t leaf_stride,
                                const uint32_t *leaf_count, uint64_t *out_pos, uint32_t *out_key,
                                uint32_t cand_stride, uint32_t *out_cand_count);
/* search_leaves (hierarchical.rs:439-464) for given leaf lists: scan + select_top_candidates
 * (:645-655) + rerank_with_full_precision (:659-674) + finalize_results (:678-691). Also
 * returns, when non-NULL, the phase-1 selection in phase-1 order: sel_pos/sel_key
 * [n_queries][sel_stride] and sel_dist (reranked distance of each). */
cuddle_status cuddle_search_leaves_batch(cuddle_index *idx, const float *queries,
                                         uint32_t n_queries, uint32_t dim, const uint32_t *leaves,
                                         uint32_t leaf_stride, const uint32_t *leaf_count,
                                         uint32_t k, uint32_t rerank_factor, uint32_t flags,
                                         uint64_t *out_ids, float *out_dist, uint32_t *out_count,
                                         uint64_t *sel_pos, uint32_t *sel_key, float *sel_dist,
                                         uint32_t sel_stride, uint32_t *sel_count);

/* ---- introspection ---------------------------------------------------------------------- */
uint32_t cuddle_index_dim(const cuddle_index *idx);
uint32_t cuddle_index_levels(const cuddle_index *idx);       /* max tree depth + 1 */
uint32_t cuddle_index_max_children(const cuddle_index *idx);
uint32_t cuddle_index_max_leaf_size(const cuddle_index *idx);
uint64_t cuddle_index_device_bytes(const cuddle_index *idx); /* index arrays + workspace */

/* ---- leaf-sharded exact search ------------------------------------------------------------
 * No counterpart in the reference, which is a single process (sharding is prose only,
 * docs/goals/SCALING_GOALS.md:139-147). One index is split by leaves over several handles
 * (one per GPU; the tree is replicated, a leaf's vectors live on exactly one shard). The three
 * calls below are cuddle_search_batch_device cut at the two points where the shards must agree,
 * so that the union of their selections is exactly select_top_candidates
 * (src/index/hierarchical.rs:645-655) of the whole index:
 *   cuddle_shard_scan   : query codes, routing, scan of the shard's leaves. d_hist[q][dim+1] gets
 *                         the shard's Hamming histogram. -> the caller sums d_hist over the shards
 *                         (all-reduce).
 *   cuddle_shard_ties   : threshold and tie budget from the summed histogram; d_tie_count
 *                         [q][*leaf_stride] gets the shard's candidates at the threshold distance
 *                         per slot of the query's leaf list. -> summed over the shards likewise.
 *   cuddle_shard_finish : the shard's members of the global selection, rerank, local top-k.
 *                         d_out_order[q][k] = each result's place in the global phase-1 order
 *                         (Hamming << 48 | leaf slot << 24 | position in the leaf).
 * The shards' lists are then merged by (distance, order): cuddle_merge_topk_ordered, or the same
 * rule on the all-gathered device buffers (vectordb_amd/sharded.py).
 * All pointers are device pointers; work is enqueued on `stream`; 1..4096 queries per batch;
 * binary codes (RaBitQ codes: cuddle_shard_scan_rabitq below). The handle keeps the batch's state
 * between the calls: no other call on it in between. */
cuddle_status cuddle_shard_scan(cuddle_index *idx, const float *d_queries, uint32_t n_queries,
                                uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                uint32_t rerank_factor, uint32_t flags, uint32_t *d_hist,
                                uint32_t *leaf_stride, void *stream);
/* Routing is the same on every shard (the tree is replicated), so the shards may split it by
 * queries instead of repeating it: each routes a slice of the batch with
 * cuddle_route_batch_device, the leaf lists are all-gathered, and every shard scans the whole
 * batch with cuddle_shard_scan_leaves (cuddle_shard_scan with the routing replaced by the given
 * lists: d_leaves [n_queries][leaf_stride_in] node ids in accumulation order, d_leaf_count
 * [n_queries]; ids beyond the tree are clamped). */
cuddle_status cuddle_shard_scan_leaves(cuddle_index *idx, const float *d_queries,
                                       uint32_t n_queries, uint32_t dim, uint32_t k,
                                       uint32_t probes_per_level, uint32_t rerank_factor,
                                       uint32_t flags, const uint32_t *d_leaves,
                                       uint32_t leaf_stride_in, const uint32_t *d_leaf_count,
                                       uint32_t *d_hist, uint32_t *leaf_stride, void *stream);
cuddle_status cuddle_shard_ties(cuddle_index *idx, const uint32_t *d_global_hist,
                                uint32_t *d_tie_count, void *stream);
cuddle_status cuddle_shard_finish(cuddle_index *idx, const float *d_queries,
                                  const uint32_t *d_global_ties, uint64_t *d_out_ids,
                                  float *d_out_dist, uint64_t *d_out_order, uint32_t *d_out_count,
                                  void *stream);

/* The same for RaBitQ codes, whose phase-1 keys are f32 estimates (hierarchical_with_rabitq.rs:605-633): every
 * shard makes the selection it would make alone (the rerank_k best of its own candidates; the selection of
 * the whole index is a subset of the union of these) and names each member by a 64-bit key that orders
 * candidates as the unsharded selection does: estimate, then leaf slot, then place in the leaf.
 *   cuddle_shard_scan_rabitq   : query preparation, routing (d_leaves == NULL) or the given leaf lists, scan,
 *                                the shard's own selection. d_keys [n_queries][R] receives its keys in
 *                                ascending order, padded with UINT64_MAX; R = cuddle_rerank_k(k, rerank_factor,
 *                                flags). -> the caller gathers the keys of all shards (all-gather).
 *   cuddle_shard_finish_rabitq : d_all_keys [n_shards][n_queries][R]; the R-th smallest key over all shards
 *                                is the threshold, the shard keeps its selection up to it, reranks, and returns
 *                                its local top-k with d_out_order = the keys.
 * The lists are merged as for binary codes (cuddle_merge_topk_ordered*). Leaf lists of fewer than 16384
 * slots, leaves of at most 262144 vectors. */
cuddle_status cuddle_rerank_k(uint32_t k, uint32_t rerank_factor, uint32_t flags, uint32_t *out);
cuddle_status cuddle_shard_scan_rabitq(cuddle_index *idx, const float *d_queries, uint32_t n_queries,
                                       uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                       uint32_t rerank_factor, uint32_t flags, const uint32_t *d_leaves,
                                       uint32_t leaf_stride_in, const uint32_t *d_leaf_count,
                                       uint64_t *d_keys, void *stream);
cuddle_status cuddle_shard_finish_rabitq(cuddle_index *idx, const float *d_queries,
                                         const uint64_t *d_all_keys, uint32_t n_shards,
                                         uint64_t *d_out_ids, float *d_out_dist, uint64_t *d_out_order,
                                         uint32_t *d_out_count, void *stream);

/* ---- multi-GPU merge -------------------------------------------------------------------- */
/* Merge `n_lists` per-shard result lists (each [n_queries][k], ascending, padded with
 * id = UINT64_MAX / +inf) into the global top-k per query, ties by (distance, list index,
 * rank in list). Host computation on host pointers; the device path does the same merge on
 * the all-gathered buffers (see vectordb_amd/sharded.py). lists_* are laid out
 * [n_lists][n_queries][k]. */
cuddle_status cuddle_merge_topk(const uint64_t *lists_ids, const float *lists_dist,
                                uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                uint64_t *out_ids, float *out_dist, uint32_t *out_count);

/* The same for the lists of cuddle_shard_finish: ties by (distance, order), which is the
 * finalize_results order (hierarchical.rs:678-691) of the unsharded index. */
cuddle_status cuddle_merge_topk_ordered(const uint64_t *lists_ids, const float *lists_dist,
                                        const uint64_t *lists_order, uint32_t n_lists,
                                        uint32_t n_queries, uint32_t k, uint64_t *out_ids,
                                        float *out_dist, uint32_t *out_count);

/* The same merge on device buffers, enqueued on `stream` (what the ranks run on the all-gathered
 * lists): n_lists * k <= 1024, else CUDDLE_ERR_UNSUPPORTED. list_stride_bytes = 0:
 * the arrays are laid out [n_lists][n_queries][k] like the host merge's. Otherwise list l of each
 * of the three arrays ([n_queries][k]) starts list_stride_bytes (a multiple of 8) after list
 * l - 1, so that a rank's ids, distances and order keys can travel as one message. */
cuddle_status cuddle_merge_topk_ordered_device(const uint64_t *d_lists_ids, const float *d_lists_dist,
                                               const uint64_t *d_lists_order, uint32_t n_lists,
                                               uint32_t n_queries, uint32_t k,
                                               uint64_t list_stride_bytes, uint64_t *d_out_ids,
                                               float *d_out_dist, uint32_t *d_out_count,
                                               void *stream);

/* ---- leaf-sharded search over the GPUs of one node (RCCL over xGMI) ---------------------------
 * No counterpart in the reference (a single process; sharding is prose only,
 * docs/goals/SCALING_GOALS.md:139-147). One process per GPU: each holds one leaf shard of the
 * index (the whole tree, the vectors of its own leaves) as a cuddle_index and one communicator.
 * Rank 0 makes the id, the host application hands it to the other ranks (its own channel: a
 * file, a socket, MPI ...), every rank calls cuddle_comm_init. RCCL is loaded on first use;
 * CUDDLE_ERR_RCCL when it is missing or a collective fails. */
#define CUDDLE_COMM_ID_BYTES 128
typedef struct cuddle_comm cuddle_comm;
cuddle_status cuddle_comm_unique_id(uint8_t *id /* [CUDDLE_COMM_ID_BYTES] */);
cuddle_status cuddle_comm_init(int rank, int world, const uint8_t *id, int device_id, cuddle_comm **out);
void cuddle_comm_destroy(cuddle_comm *comm);
int cuddle_comm_rank(const cuddle_comm *comm);
int cuddle_comm_world(const cuddle_comm *comm);
/* cuddle_search_batch_device of the WHOLE index, computed by all ranks together (every rank calls
 * it with the same queries and parameters and receives the same result, which is the unsharded
 * index's result bit for bit): routing split by queries + all-gather of the leaf lists,
 * cuddle_shard_scan_leaves + all-reduce of the Hamming histograms, cuddle_shard_ties +
 * all-reduce of the tie counts, cuddle_shard_finish + all-gather of the local top-k,
 * cuddle_merge_topk_ordered_device. With RaBitQ codes the two all-reduces are replaced by one all-gather
 * of the shards' selection keys (cuddle_shard_scan_rabitq / cuddle_shard_finish_rabitq). Everything is
 * enqueued on `stream`. Batches of more than 4096 queries run as equal rounds of at most 4096, one after the
 * other; world * k <= 1024. */
cuddle_status cuddle_sharded_search_batch(cuddle_index *shard, cuddle_comm *comm, const float *d_queries,
                                          uint32_t n_queries, uint32_t dim, uint32_t k,
                                          uint32_t probes_per_level, uint32_t rerank_factor,
                                          uint32_t flags, uint64_t *d_out_ids, float *d_out_dist,
                                          uint32_t *d_out_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUDDLE_HIP_H */
