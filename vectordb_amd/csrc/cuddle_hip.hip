// cuddle_hip.hip — host side + C ABI of libcuddle_hip.so (see include/cuddle_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// The library is self-contained HIP: no torch types, no CPU fallback. Without a visible GPU
// every compute entry point returns CUDDLE_ERR_NO_DEVICE.
#include "../../include/cuddle_hip.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <climits>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "cuddle_kernels.hpp"
#include "cuddle_route.hpp"
#include "cuddle_shard.hpp"
#include "cuddle_rabitq.hpp"
#include "cuddle_rabitq_grouped.hpp"
#include "cuddle_build.hpp"

using namespace cuddle;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static cuddle_status fail(cuddle_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? CUDDLE_ERR_OOM : CUDDLE_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                           \
    } while (0)

#define ST_TRY(expr)                        \
    do {                                    \
        cuddle_status s__ = (expr);         \
        if (s__ != CUDDLE_OK) return s__;   \
    } while (0)

// Two HIP runtimes in one process. PyTorch wheels carry their own libamdhip64 and load it by file name; this library
// is linked against the system's. When this library is loaded first, both end up mapped, each with its own view of
// the devices, and the second one to start reports "no ROCm-capable device" - far from the cause. The library
// therefore refuses to create an index in such a process and names both files (CUDDLE_ALLOW_TWO_HIP_RUNTIMES=1
// overrides, for hosts that never touch the GPU through the other copy).
static cuddle_status check_hip_runtime() {
    if (getenv("CUDDLE_ALLOW_TWO_HIP_RUNTIMES")) return CUDDLE_OK;
    Dl_info info{};
    char mine[PATH_MAX] = "", other[PATH_MAX], line[PATH_MAX + 128];
    if (!dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) || !info.dli_fname || !realpath(info.dli_fname, mine))
        return CUDDLE_OK; // cannot tell which runtime this library is bound to
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return CUDDLE_OK;
    cuddle_status st = CUDDLE_OK;
    while (st == CUDDLE_OK && fgets(line, sizeof(line), f)) {
        const char *path = strchr(line, '/');
        if (!path || !strstr(path, "/libamdhip64.so")) continue;
        std::string p(path);
        while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
        if (!realpath(p.c_str(), other) || strcmp(other, mine) == 0) continue;
        st = fail(CUDDLE_ERR_HIP,
                  "two HIP runtimes are loaded in this process: libcuddle_hip.so is bound to %s, and %s is mapped as well. "
                  "Load the other user of HIP first (Python: `import torch` before vectordb_amd), so that one runtime serves both",
                  mine, other);
    }
    fclose(f);
    return st;
}

static cuddle_status use_device(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(CUDDLE_ERR_NO_DEVICE, "no HIP device visible");
    }
    if (device_id < 0 || device_id >= n) return fail(CUDDLE_ERR_BAD_ARG, "device_id %d out of range (0..%d)", device_id, n - 1);
    HIP_TRY(hipSetDevice(device_id));
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// fixed work space of cuddle_search_batch_device_noalloc
// ---------------------------------------------------------------------------------------------
// Every GPU the library runs on holds its own copy of this array (at its own address) until the process
// ends. In host code the name is only a handle for hipGetSymbolAddress: the device address is looked up
// once per device (scratch_for_device, called with that device current) and everything on the host side
// - kernel arguments, memsets - uses that address, never the name.
static constexpr size_t kScratchBytes = 32u << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

static constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[kMaxDevices];          // device address of `scratch` on each device, once resolved
static hipEvent_t g_scratch_ev[kMaxDevices];      // orders a call behind the previous user's stream (created with the address)
static std::mutex g_scratch_call_mu[kMaxDevices]; // held across a whole enqueue that uses the device's array
static hipStream_t g_scratch_stream[kMaxDevices]; // stream of the last call that used it
static bool g_scratch_used[kMaxDevices];

// Device address of `scratch` on `device_id`, which must be the current device.
static cuddle_status scratch_for_device(int device_id, void **out) {
    if (device_id < 0 || device_id >= kMaxDevices) return fail(CUDDLE_ERR_UNSUPPORTED, "device %d: at most %d devices", device_id, kMaxDevices);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device_id]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hipGetLastError();
            return fail(CUDDLE_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device_id, hipGetErrorString(e));
        }
        e = hipEventCreateWithFlags(&g_scratch_ev[device_id], hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(CUDDLE_ERR_HIP, "hipEventCreate failed on device %d: %s", device_id, hipGetErrorString(e));
        }
        g_scratch_ptr[device_id] = p;
    }
    *out = g_scratch_ptr[device_id];
    return CUDDLE_OK;
}

// Is `s` recording a graph? Calls that would wait, record events for later calls or copy to the host are left out then.
static bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// Orders stream `s` behind everything enqueued so far on `prev`, on the device: no host wait. If `prev` no longer
// exists (its owner destroyed it) the whole device is waited for instead, once, and the call goes on.
static void order_after(hipStream_t prev, hipStream_t s, hipEvent_t ev) {
    if (ev && hipEventRecord(ev, prev) == hipSuccess && hipStreamWaitEvent(s, ev, 0) == hipSuccess) return;
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    cuddle_status ensure(size_t need) {
        if (need <= bytes) return CUDDLE_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t alloc = (need + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&p, alloc);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(CUDDLE_ERR_OOM, "hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e));
        }
        bytes = alloc;
        return CUDDLE_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct DevStats { // accumulated on device by stats_accum_kernel
    unsigned long long leaves, scanned, reranked;
    uint32_t max_cand, max_leaves, fallbacks, pad1;
    unsigned long long rq_exact, rq_exact_all; // RaBitQ scan: candidates evaluated exactly, queries that evaluated all of theirs
    unsigned long long rq_mismatch;            // ... bounds of the grouped kernel that differ from the per-query ones (checking mode)
};

// Work-space layout of one chunk of queries.
struct Plan {
    uint32_t qc = 0;        // queries per chunk
    uint32_t k = 0, probes = 0;
    uint32_t rerank_k = 0;
    uint32_t fstride = 0;   // frontier capacity per query
    uint32_t lstride = 0;   // leaf-list capacity per query
    uint32_t cstride = 0;   // candidate capacity per query (spill buffer)
    uint32_t sstride = 0;   // selection capacity per query
    uint32_t dist_cap = 0;  // candidates kept in LDS by the scan kernel
    uint32_t *loc_key = nullptr, *rr_order = nullptr; // [qc] locality key of each query; the queries in key order (the rerank's order)
    uint32_t tcap = 0;      // 64-vector tiles per query (scan kernel)
    uint32_t scan_threads = 512; // block size of the scan kernels
    size_t scan_lds = 0;
    // carved pointers
    uint64_t *qcodes = nullptr;
    uint32_t *frontier[2] = {nullptr, nullptr}, *fcount[2] = {nullptr, nullptr};
    float *fdist = nullptr;
    uint32_t *leaves = nullptr, *lcount = nullptr;
    uint16_t *gdist = nullptr;
    // (query, parent) pairs of the grouped routing level
    uint32_t *pair_info = nullptr, *pair_parent = nullptr, *pair_slot = nullptr, *pair_sorted = nullptr, *small_flag = nullptr;
    uint32_t *seg[2] = {nullptr, nullptr}; // segment tables standing for per-query frontiers (see PairOut::seg), like frontier[]
    uint32_t *grp_cnt = nullptr, *grp_poff = nullptr, *grp_ioff = nullptr, *grp_fill = nullptr, *grp_item = nullptr;
    // leaf-sharded exact selection: threshold/budget per query, global order key of every selected candidate
    uint32_t *thr = nullptr;
    uint64_t *sel_ord = nullptr;
    float *qnorm = nullptr; // cosine: sqrt(dot(q, q)) per query
    uint32_t *rq_rows = nullptr; // RaBitQ: [qc][rq_plane_stride] quantised query magnitudes as bit planes (rabitq_query_planes_kernel)
    // RaBitQ bounds on the matrix cores (rq_bounds_grouped_kernel): the queries' signed pieces, the (query, leaf) pairs
    // of the leaf lists in leaf order, the work items cut from them
    int8_t *rq_i8 = nullptr;
    uint32_t *rq_pinfo = nullptr, *rq_pnode = nullptr, *rq_pslot = nullptr, *rq_psorted = nullptr;
    uint32_t *rq_cnt = nullptr, *rq_poff = nullptr, *rq_ioff = nullptr, *rq_bsum = nullptr, *rq_items = nullptr, *rq_order = nullptr;
    bool rq_grouped = false;
    bool leaf_grouped = false; // binary codes: the Hamming scan leaf by leaf (leaf_hamming_kernel), same pair arrays
    uint32_t *level_fc = nullptr; // [n_levels + 1][qc] frontier size of every query at every level (stats variant)
    uint32_t *leaf_tables = nullptr; // [qc][4][lstride + 4] per-slot tables of the reached leaves (cand_count_kernel)
    uint32_t *sel_pos = nullptr, *sel_key = nullptr, *sel_count = nullptr, *cand_count = nullptr;
    float *sel_dist = nullptr;
    // MFMA routing
    uint32_t kc_cap = 0;    // prefilter candidates per query (probes + margin)
    uint32_t kc_split = 0;  // ... of a level routed with the split-bf16 filter (<= kc_cap)
    uint32_t kc_half = 0;   // ... of a level routed with the single-pass half-precision filter (<= kc_cap)
    float *qsq = nullptr, *rc_dist = nullptr, *rc_max = nullptr;
    float *qinv = nullptr;     // inverse scale of every query's half-precision row
    uint16_t *qcent = nullptr; // centred queries as bf16 (hi, lo) pairs, or as scaled fp16
    uint32_t *rc_pos = nullptr, *rflag = nullptr, *rc_cnt = nullptr;
    size_t bytes = 0;
};

// Per-level operand of the MFMA routing contraction (see level_table_kernel).
struct LevelGemm {
    uint32_t n = 0;            // 0: contraction operands not built for this level
    uint32_t *nodes = nullptr; // [level width] node ids, breadth-first order (every level has it)
    uint2 *child_rows = nullptr; // [level width] (position of the node's first child in the next level, children)
    uint16_t *table = nullptr; // [n][split_kp(dim)] centroid - mu as bf16 (hi, lo) pairs, see level_table_kernel
    float *sqn = nullptr;      // [n]
    float *mu = nullptr;       // [dim]
    float max_sqn = 0.f;
    // single-pass half-precision operands (level_table16_kernel), built for levels that can be a whole frontier
    uint16_t *table16 = nullptr; // [n][split_kp(dim)] scaled fp16
    float *inv16 = nullptr;      // [n] inverse scales
};

struct cuddle_index {
    int device = 0;
    IndexDev dev{};
    std::vector<void *> owned; // device allocations of the index arrays
    uint64_t index_bytes = 0;
    uint32_t n_levels = 0, max_children = 0, max_leaf = 0, n_leaves = 0;
    float threshold = 0.f; // BinaryQuantizer::threshold
    uint32_t n_cus = 256;
    std::vector<uint32_t> level_width;
    std::vector<LevelGemm> lvl;
    std::vector<uint32_t> level_leaves; // leaf nodes per level
    uint32_t max_parent_width = 1;      // widest level that has a level below it
    DevBuf ws;       // per-chunk work space (Plan)
    DevBuf io;       // staging for the host entry points
    DevStats *d_stats = nullptr;
    void *scratch_ptr = nullptr; // this device's copy of the library's fixed work space
    std::mutex mu;
    // stage timing of the last timed search
    std::vector<hipEvent_t> events;
    std::vector<int> event_stage; // stage whose work follows the event on the stream, or -1
    size_t events_used = 0;
    bool timed = false;
    uint32_t timed_searches = 0;
    hipStream_t last_stream = nullptr;
    bool stream_used = false; // last_stream names the stream of an earlier call
    hipEvent_t ws_ev = nullptr; // orders a call on another stream behind the previous call's stream (enter_stream)
    // Two lanes: a batch is cut into pieces that run on two streams of the handle's own, each with its half of the
    // work space, so that one piece's short dependent kernels fill the gaps of the other's long ones (search_device_locked).
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, lane_ev[2] = {nullptr, nullptr};
    // The single-pass half-precision routing filter is judged per level by how often its proof fails: a level whose
    // queries fall back to the exact kernels too often goes back to the split filter for good (HalfFeedback below).
    std::vector<uint8_t> half_ok;      // [n_levels] the level may use the half-precision filter
    std::vector<uint32_t> half_seen;   // [n_levels] failures already judged
    uint32_t *d_half_fb = nullptr;     // [n_levels] failures counted by route_finish_kernel, never reset
    uint32_t *h_half_fb = nullptr;     // pinned snapshot of d_half_fb
    hipEvent_t half_ev = nullptr;      // the snapshot's copy has completed
    bool half_pending = false;
    uint64_t half_q_total = 0, half_q_posted = 0, half_q_seen = 0; // queries routed through a half-filtered level
    uint32_t half_calls = 0;
    // The per-query counts of a single-chunk search stay in the work space until the handle's next call; their sums
    // (leaves, candidates, selections) are formed when cuddle_stats asks, not by a kernel behind every search.
    const uint32_t *lazy_lcount = nullptr, *lazy_cand = nullptr, *lazy_sel = nullptr;
    uint32_t lazy_nq = 0;
    bool lazy_pending = false;
    uint32_t last_levels = 0;
    // leaf-sharded exact search: the plan of the batch between cuddle_shard_scan and cuddle_shard_finish
    Plan shard_plan;
    uint32_t shard_nq = 0;
    int shard_state = 0; // 0 idle, 1 scanned, 2 ties counted
    ~cuddle_index() {
        for (void *p : owned) (void)hipFree(p);
        ws.release();
        io.release();
        if (d_stats) (void)hipFree(d_stats);
        if (d_half_fb) (void)hipFree(d_half_fb);
        if (h_half_fb) (void)hipHostFree(h_half_fb);
        if (half_ev) (void)hipEventDestroy(half_ev);
        if (ws_ev) (void)hipEventDestroy(ws_ev);
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        for (int l = 0; l < 2; ++l) {
            if (lane_ev[l]) (void)hipEventDestroy(lane_ev[l]);
            if (lane_stream[l]) (void)hipStreamDestroy(lane_stream[l]);
        }
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

static constexpr uint32_t kMaxChunk = 4096;
static constexpr size_t kWorkspaceBudget = 3ull << 30; // the handle's work space: chunks shrink before it grows beyond this
static constexpr size_t kMaxEvents = 1u << 18; // stage-timing events kept between two cuddle_stage_times() calls
static constexpr uint32_t kScanThreads = 512;
// Split-table size above which a level of per-query frontiers is contracted from half-precision operands: the
// 256 MiB Infinity Cache (CUDDLE_GROUPED_HALF_MIN_BYTES overrides it: tests force the path on small trees).
static uint64_t grouped_half_min_bytes() {
    const char *e = getenv("CUDDLE_GROUPED_HALF_MIN_BYTES"); // (read at every call: a test sets it for one index)
    return e ? strtoull(e, nullptr, 10) : (256ull << 20);
}

static constexpr uint32_t kGemmMinWidth = 512; // narrower levels are routed by the exact kernel
static constexpr uint32_t kMaxLeafVectors = 4096u * 64u; // the scan's tile tables hold the tile number inside a leaf in 12 bits
static constexpr uint32_t kMaxCandidates = 1u << 25;    // ... and a tile's scan position in 25 bits
static constexpr uint32_t kLdsDistCap = 24576; // u16 distances kept in LDS per query (48 KiB)

using GroupedKernel = void (*)(IndexDev, const float *, const uint32_t *, uint32_t, const uint32_t *, const uint4 *,
                               const uint32_t *, float *, uint32_t, const float *);

template <int NBLK>
static GroupedKernel grouped_kernel_metric(uint32_t metric) {
    if (metric == CUDDLE_METRIC_L2) return route_dist_grouped_kernel<NBLK, kMetricL2>;
    if (metric == CUDDLE_METRIC_DOT) return route_dist_grouped_kernel<NBLK, kMetricDot>;
    return route_dist_grouped_kernel<NBLK, kMetricCosine>;
}

// The grouped routing kernel for an index: centroid slices held in registers for the common
// dimensions, the any-dimension variant otherwise.
static GroupedKernel grouped_kernel_for(uint32_t dim, uint32_t metric) {
    switch (dim) {
    case 64: return grouped_kernel_metric<4>(metric);
    case 96: return grouped_kernel_metric<6>(metric);
    case 128: return grouped_kernel_metric<8>(metric);
    case 256: return grouped_kernel_metric<16>(metric);
    default: return grouped_kernel_metric<0>(metric);
    }
}

#ifndef CUDDLE_SCAN_U2
#define CUDDLE_SCAN_U2 8 // tiles a wave keeps in flight when a code is one 16-byte slot (dim <= 128)
#endif
using ScanKernel = void (*)(IndexDev, const uint64_t *, const uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                            void *, uint32_t, uint32_t *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *);

// The Hamming scan instantiation for a code of `words` u64 (distances in u8 when dim < 255).
static ScanKernel scan_kernel_for(uint32_t dim, uint32_t words) {
    switch (words) {
    case 2: return dim < 255 ? scan_select_hamming_kernel<uint8_t, 1, false, CUDDLE_SCAN_U2> : scan_select_hamming_kernel<uint16_t, 1, false, CUDDLE_SCAN_U2>;
    case 4: return dim < 255 ? scan_select_hamming_kernel<uint8_t, 2, false, 8> : scan_select_hamming_kernel<uint16_t, 2, false, 8>;
    case 8: return scan_select_hamming_kernel<uint16_t, 4, false, 4>;   // 512-d
    case 12: return scan_select_hamming_kernel<uint16_t, 6, false, 2>;  // 768-d
    case 15: return scan_select_hamming_kernel<uint16_t, 7, true, 2>;   // 960-d
    case 16: return scan_select_hamming_kernel<uint16_t, 8, false, 2>;  // 1024-d
    case 24: return scan_select_hamming_kernel<uint16_t, 12, false, 1>; // 1536-d
    default: return dim < 255 ? scan_select_hamming_kernel<uint8_t, 0, false, 1> : scan_select_hamming_kernel<uint16_t, 0, false, 1>;
    }
}

using CeilingKernel = void (*)(IndexDev, const uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t *);

// The read-only twin of scan_kernel_for()'s instantiation (same PAIRS / TAIL / U); null for the any-width kernel.
static CeilingKernel ceiling_kernel_for(uint32_t words) {
    switch (words) {
    case 2: return scan_read_ceiling_kernel<1, false, CUDDLE_SCAN_U2>;
    case 4: return scan_read_ceiling_kernel<2, false, 8>;
    case 8: return scan_read_ceiling_kernel<4, false, 4>;
    case 12: return scan_read_ceiling_kernel<6, false, 2>;
    case 15: return scan_read_ceiling_kernel<7, true, 2>;
    case 16: return scan_read_ceiling_kernel<8, false, 2>;
    case 24: return scan_read_ceiling_kernel<12, false, 1>;
    default: return nullptr;
    }
}

template <class T>
static cuddle_status upload(cuddle_index *ix, const T *src, size_t count, T **dst) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(CUDDLE_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ix->owned.push_back(p);
    ix->index_bytes += bytes;
    if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyDefault)); // host or device source
    *dst = static_cast<T *>(p);
    return CUDDLE_OK;
}

template <class T>
static cuddle_status dev_alloc(cuddle_index *ix, size_t count, T **dst) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(CUDDLE_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ix->owned.push_back(p);
    ix->index_bytes += bytes;
    *dst = static_cast<T *>(p);
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// small kernels that live with the host code
// ---------------------------------------------------------------------------------------------
namespace cuddle {
__global__ void stats_accum_kernel(const uint32_t *lcount, const uint32_t *cand_count, const uint32_t *sel_count,
                                   uint32_t nq, DevStats *st) {
    unsigned long long l = 0, c = 0, r = 0;
    uint32_t mc = 0, ml = 0;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        l += lcount[q]; c += cand_count[q]; r += sel_count[q];
        mc = max(mc, cand_count[q]); ml = max(ml, lcount[q]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        l += __shfl_xor(l, o, 64); c += __shfl_xor(c, o, 64); r += __shfl_xor(r, o, 64);
        mc = max(mc, (uint32_t)__shfl_xor((int)mc, o, 64)); ml = max(ml, (uint32_t)__shfl_xor((int)ml, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&st->leaves, l);
        atomicAdd(&st->scanned, c);
        atomicAdd(&st->reranked, r);
        atomicMax(&st->max_cand, mc);
        atomicMax(&st->max_leaves, ml);
    }
}
// Sum of squares of each centroid as a k-ordered fused-multiply-add chain, used by the MFMA routing filter.
__global__ void centroid_sqnorm_kernel(const float *c, uint32_t n, uint32_t dim, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (uint32_t j = 0; j < dim; ++j) s = __builtin_fmaf(c[(uint64_t)i * dim + j], c[(uint64_t)i * dim + j], s);
    out[i] = s;
}
} // namespace cuddle

// ---------------------------------------------------------------------------------------------
// library
// ---------------------------------------------------------------------------------------------
extern "C" uint32_t cuddle_abi_version(void) { return CUDDLE_ABI_VERSION; }
extern "C" const char *cuddle_last_error(void) { return g_last_error.c_str(); }
extern "C" int cuddle_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// index creation
// ---------------------------------------------------------------------------------------------
static cuddle_status validate_desc(const cuddle_index_desc *d, std::vector<uint32_t> &depth, uint32_t &n_levels,
                                   std::vector<uint32_t> &level_width, std::vector<std::vector<uint32_t>> &level_nodes) {
    if (!d) return fail(CUDDLE_ERR_BAD_ARG, "desc is null");
    if (d->abi_version != CUDDLE_ABI_VERSION) return fail(CUDDLE_ERR_BAD_ARG, "abi_version %u != %u", d->abi_version, CUDDLE_ABI_VERSION);
    if (d->dim == 0 || d->dim > 32768) return fail(CUDDLE_ERR_UNSUPPORTED, "dim %u outside 1..32768", d->dim);
    if (d->metric > CUDDLE_METRIC_DOT) return fail(CUDDLE_ERR_BAD_ARG, "unknown metric %u", d->metric);
    if (d->code_kind > CUDDLE_CODE_RABITQ) return fail(CUDDLE_ERR_BAD_ARG, "unknown code_kind %u", d->code_kind);
    if (d->n_nodes == 0 || d->n_roots == 0 || d->n_roots > d->n_nodes) return fail(CUDDLE_ERR_BAD_INDEX, "n_nodes/n_roots invalid");
    if (d->n_vectors >= (1ull << 32)) return fail(CUDDLE_ERR_UNSUPPORTED, "n_vectors >= 2^32 per shard");
    if (!d->centroids || !d->child_offsets || !d->roots || !d->vec_offsets) return fail(CUDDLE_ERR_BAD_ARG, "null tree array");
    if (d->n_vectors && (!d->ids || !d->vectors)) return fail(CUDDLE_ERR_BAD_ARG, "null ids/vectors");
    if (d->code_kind == CUDDLE_CODE_BINARY && d->n_vectors && !d->codes) return fail(CUDDLE_ERR_BAD_ARG, "null codes");
    if (d->code_kind == CUDDLE_CODE_RABITQ && (!d->rabitq_rot_t || (d->n_vectors && (!d->rabitq_codes || !d->rabitq_norm || !d->rabitq_ip_oo))))
        return fail(CUDDLE_ERR_BAD_ARG, "null RaBitQ array");
    if (d->code_kind == CUDDLE_CODE_RABITQ && (size_t)kPrepQ * d->dim * sizeof(float) > 144 * 1024)
        return fail(CUDDLE_ERR_UNSUPPORTED, "RaBitQ codes: dim %u > %u (the query rotation stages %d queries in LDS)", d->dim,
                    (unsigned)(144 * 1024 / (kPrepQ * sizeof(float))), kPrepQ);
    const uint32_t n = d->n_nodes;
    if (d->child_offsets[0] != 0 || d->vec_offsets[0] != 0) return fail(CUDDLE_ERR_BAD_INDEX, "offsets must start at 0");
    for (uint32_t i = 0; i < n; ++i) {
        if (d->child_offsets[i + 1] < d->child_offsets[i]) return fail(CUDDLE_ERR_BAD_INDEX, "child_offsets not monotone at %u", i);
        if (d->vec_offsets[i + 1] < d->vec_offsets[i]) return fail(CUDDLE_ERR_BAD_INDEX, "vec_offsets not monotone at %u", i);
        if (d->child_offsets[i + 1] > d->child_offsets[i] && d->vec_offsets[i + 1] > d->vec_offsets[i])
            return fail(CUDDLE_ERR_BAD_INDEX, "node %u has both children and vectors", i);
    }
    if (d->vec_offsets[n] != d->n_vectors) return fail(CUDDLE_ERR_BAD_INDEX, "vec_offsets[n_nodes] != n_vectors");
    if (d->child_offsets[n] && !d->children) return fail(CUDDLE_ERR_BAD_ARG, "null children");
    // tree check + depths (breadth first from the roots)
    depth.assign(n, UINT32_MAX);
    std::vector<uint32_t> cur(d->roots, d->roots + d->n_roots), nxt;
    uint32_t lvl = 0;
    level_width.clear();
    level_nodes.clear();
    while (!cur.empty()) {
        level_width.push_back((uint32_t)cur.size());
        level_nodes.push_back(cur);
        nxt.clear();
        for (uint32_t v : cur) {
            if (v >= n) return fail(CUDDLE_ERR_BAD_INDEX, "node id %u out of range", v);
            if (depth[v] != UINT32_MAX) return fail(CUDDLE_ERR_BAD_INDEX, "node %u reachable twice (not a forest)", v);
            depth[v] = lvl;
            for (uint32_t c = d->child_offsets[v]; c < d->child_offsets[v + 1]; ++c) nxt.push_back(d->children[c]);
        }
        cur.swap(nxt);
        if (++lvl > 64) return fail(CUDDLE_ERR_BAD_INDEX, "tree deeper than 64 levels");
    }
    n_levels = lvl;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_index_create(const cuddle_index_desc *d, int device_id, cuddle_index **out) {
    if (!out) return fail(CUDDLE_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    std::vector<uint32_t> depth, level_width;
    std::vector<std::vector<uint32_t>> level_nodes;
    uint32_t n_levels = 0;
    ST_TRY(validate_desc(d, depth, n_levels, level_width, level_nodes));
    ST_TRY(check_hip_runtime());
    ST_TRY(use_device(device_id));
    cuddle_index *ix = new (std::nothrow) cuddle_index();
    if (!ix) return fail(CUDDLE_ERR_OOM, "host allocation failed");
    struct Guard {
        cuddle_index *p;
        ~Guard() { delete p; }
    } guard{ix};
    ix->device = device_id;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ix->n_cus = (uint32_t)prop.multiProcessorCount;
    }
    ix->threshold = d->threshold;
    ix->n_levels = n_levels;
    ix->level_width = level_width;
    const uint32_t n = d->n_nodes, D = d->dim, W = (D + 63) / 64, W32 = (D + 31) / 32;
    const uint64_t N = d->n_vectors;
    IndexDev &v = ix->dev;
    v.dim = D; v.words = W; v.words32 = W32; v.metric = d->metric; v.code_kind = d->code_kind;
    v.n_nodes = n; v.n_roots = d->n_roots; v.n_vectors = N;

    // host-side derived arrays
    std::vector<uint32_t> vec_off32(n + 1), code_off(n, 0);
    const bool dev_arrays = (d->flags & CUDDLE_DESC_DEVICE_ARRAYS) != 0;
    const bool adopt = dev_arrays && (d->flags & CUDDLE_DESC_ADOPT_VECTORS) != 0;
    uint64_t code_units = 0; // 16-byte units
    for (uint32_t i = 0; i <= n; ++i) vec_off32[i] = (uint32_t)d->vec_offsets[i];
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t cnt = vec_off32[i + 1] - vec_off32[i];
        uint32_t cc = d->child_offsets[i + 1] - d->child_offsets[i];
        ix->max_children = std::max(ix->max_children, cc);
        if (cc == 0) {
            ix->n_leaves++;
            ix->max_leaf = std::max(ix->max_leaf, cnt);
        }
        if (cnt > kMaxLeafVectors)
            return fail(CUDDLE_ERR_UNSUPPORTED, "leaf %u holds %u vectors; at most %u per leaf (split the leaf)", i, cnt, kMaxLeafVectors);
        code_off[i] = (uint32_t)code_units;
        code_units += (uint64_t)(W / 2) * cnt + ((W & 1) ? (cnt + 1) / 2 : 0);
        if (code_units >= (1ull << 32)) return fail(CUDDLE_ERR_UNSUPPORTED, "binary codes exceed 64 GiB per shard");
    }

    float *centroids; uint32_t *child_off, *children, *roots, *vec_off, *coff; uint64_t *ids; float *vectors;
    ST_TRY(upload(ix, d->centroids, (size_t)n * D, &centroids));
    ST_TRY(upload(ix, d->child_offsets, (size_t)n + 1, &child_off));
    ST_TRY(upload(ix, d->children, (size_t)d->child_offsets[n], &children));
    ST_TRY(upload(ix, d->roots, (size_t)d->n_roots, &roots));
    ST_TRY(upload(ix, vec_off32.data(), (size_t)n + 1, &vec_off));
    ST_TRY(upload(ix, code_off.data(), (size_t)n, &coff));
    if (adopt) { // used in place; the caller keeps them alive
        ids = const_cast<uint64_t *>(d->ids);
        vectors = const_cast<float *>(d->vectors);
    } else {
        ST_TRY(upload(ix, d->ids, (size_t)N, &ids));
        ST_TRY(upload(ix, d->vectors, (size_t)N * D, &vectors));
    }
    v.centroids = centroids; v.child_off = child_off; v.children = children; v.roots = roots;
    v.vec_off = vec_off; v.code_off = coff; v.ids = ids; v.vectors = vectors;

    float *sqn;
    ST_TRY(dev_alloc(ix, (size_t)n, &sqn));
    hipLaunchKernelGGL(centroid_sqnorm_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, centroids, n, D, sqn);
    v.centroid_sqnorm = sqn;

    // operands of the MFMA routing contraction, for every level wide enough to use it
    ix->lvl.assign(n_levels, LevelGemm());
    ix->level_leaves.assign(n_levels, 0);
    {
        std::vector<uint32_t> level_pos(n, 0);
        for (uint32_t l = 0; l < n_levels; ++l) {
            for (size_t i = 0; i < level_nodes[l].size(); ++i) {
                const uint32_t v2 = level_nodes[l][i];
                level_pos[v2] = (uint32_t)i;
                if (d->child_offsets[v2 + 1] == d->child_offsets[v2]) ix->level_leaves[l]++;
            }
            if (l + 1 < n_levels) // (in pair counters: a narrow level has several per parent)
                ix->max_parent_width = std::max(ix->max_parent_width, (uint32_t)level_nodes[l].size() << pair_shift_of((uint32_t)level_nodes[l].size()));
            ST_TRY(upload(ix, level_nodes[l].data(), level_nodes[l].size(), &ix->lvl[l].nodes));
        }
        uint32_t *lp;
        ST_TRY(upload(ix, level_pos.data(), (size_t)n, &lp));
        v.level_pos = lp;
        std::vector<uint2> rows;
        for (uint32_t l = 0; l + 1 < n_levels; ++l) { // children of one node are consecutive in the next level
            rows.assign(level_nodes[l].size(), make_uint2(0u, 0u));
            for (size_t i = 0; i < level_nodes[l].size(); ++i) {
                const uint32_t v2 = level_nodes[l][i], cb = d->child_offsets[v2], cc = d->child_offsets[v2 + 1] - cb;
                rows[i] = make_uint2(cc ? level_pos[d->children[cb]] : 0u, cc);
            }
            ST_TRY(upload(ix, rows.data(), rows.size(), &ix->lvl[l].child_rows));
        }
    }
    { // operands of the MFMA contraction for every level wide enough to use it
        for (uint32_t l = 0; l < n_levels; ++l) {
            const std::vector<uint32_t> &nodes = level_nodes[l];
            if (nodes.size() < kGemmMinWidth) continue;
            LevelGemm &g = ix->lvl[l];
            ST_TRY(dev_alloc(ix, (size_t)D, &g.mu));
            HIP_TRY(hipMemset(g.mu, 0, D * sizeof(float)));
            if (d->metric == CUDDLE_METRIC_L2) { // mu = mean centroid of the level, accumulated in double on the device
                double *acc = nullptr;
                HIP_TRY(hipMalloc((void **)&acc, D * sizeof(double)));
                (void)hipMemset(acc, 0, D * sizeof(double));
                hipLaunchKernelGGL(level_sum_kernel, dim3((D + 63) / 64, 256), dim3(64), 0, 0, centroids, g.nodes,
                                   (uint32_t)nodes.size(), D, acc);
                hipLaunchKernelGGL(level_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, 0, acc, (uint32_t)nodes.size(), D, g.mu);
                (void)hipDeviceSynchronize();
                (void)hipFree(acc);
            }
            ST_TRY(dev_alloc(ix, nodes.size() * (size_t)split_kp(D) * 2, &g.table));
            ST_TRY(dev_alloc(ix, nodes.size(), &g.sqn));
            hipLaunchKernelGGL(level_table_kernel, dim3(((uint32_t)nodes.size() + 3) / 4), dim3(256), 0, 0, centroids,
                               g.nodes, (uint32_t)nodes.size(), D, split_kp(D), g.mu, g.table, g.sqn);
            std::vector<float> h(nodes.size());
            HIP_TRY(hipMemcpy(h.data(), g.sqn, nodes.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (float x : h) g.max_sqn = std::max(g.max_sqn, x);
            g.n = (uint32_t)nodes.size();
            // every query sees the whole level only if the level above is no wider than probes (and a level wider than
            // 4096 takes at most 4096 probes); a level of per-query frontiers gets them when its split table is too large
            // for the Infinity Cache
            if (l == 0 || level_nodes[l - 1].size() <= 4096 || nodes.size() * (uint64_t)split_kp(D) * 4 > grouped_half_min_bytes()) {
                ST_TRY(dev_alloc(ix, nodes.size() * (size_t)split_kp(D), &g.table16));
                ST_TRY(dev_alloc(ix, nodes.size(), &g.inv16));
                hipLaunchKernelGGL(level_table16_kernel, dim3(((uint32_t)nodes.size() + 3) / 4), dim3(256), 0, 0, centroids,
                                   g.nodes, (uint32_t)nodes.size(), D, split_kp(D), g.mu, reinterpret_cast<_Float16 *>(g.table16), g.inv16);
            }
        }
    }

    uint32_t *d_pos_leaf = nullptr;
    if (N) {
        HIP_TRY(hipMalloc((void **)&d_pos_leaf, N * sizeof(uint32_t)));
        hipLaunchKernelGGL(pos_leaf_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, vec_off, n, N, d_pos_leaf);
    }
    cuddle_status st = CUDDLE_OK;
    if (d->code_kind == CUDDLE_CODE_BINARY) {
        uint64_t *planar = nullptr, *raw = nullptr;
        st = dev_alloc(ix, (size_t)code_units * 2, &planar);
        if (st == CUDDLE_OK && N) {
            const uint64_t *src = d->codes; // device arrays are re-laid straight from the caller's buffer
            (void)hipMemset(planar, 0, (size_t)code_units * 16);
            if (!dev_arrays) {
                hipError_t e = hipMalloc((void **)&raw, N * W * sizeof(uint64_t));
                if (e != hipSuccess) st = fail(CUDDLE_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e));
                if (st == CUDDLE_OK) {
                    e = hipMemcpy(raw, d->codes, N * W * sizeof(uint64_t), hipMemcpyHostToDevice);
                    if (e != hipSuccess) st = fail(CUDDLE_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
                }
                src = raw;
            }
            if (st == CUDDLE_OK)
                hipLaunchKernelGGL(relayout_codes_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, src, planar,
                                   vec_off, coff, d_pos_leaf, N, W);
            (void)hipDeviceSynchronize();
            if (raw) (void)hipFree(raw);
        }
        v.codes = reinterpret_cast<const uint4 *>(planar);
    } else {
        float *rot, *rot_tr = nullptr, *norm, *ipoo; uint32_t *rc = nullptr, *raw = nullptr, *rvm = nullptr;
        st = upload(ix, d->rabitq_rot_t, (size_t)D * D, &rot);
        if (st == CUDDLE_OK) st = dev_alloc(ix, (size_t)D * D, &rot_tr);
        if (st == CUDDLE_OK)
            hipLaunchKernelGGL(transpose_square_kernel, dim3((D + 255) / 256, D), dim3(256), 0, 0, rot, rot_tr, D);
        if (st == CUDDLE_OK) st = upload(ix, d->rabitq_norm, (size_t)N, &norm);
        if (st == CUDDLE_OK) st = upload(ix, d->rabitq_ip_oo, (size_t)N, &ipoo);
        if (st == CUDDLE_OK) st = dev_alloc(ix, (size_t)N * W32, &rc);
        if (st == CUDDLE_OK && N) {
            const uint32_t *src = d->rabitq_codes;
            if (!dev_arrays) {
                hipError_t e = hipMalloc((void **)&raw, N * W32 * sizeof(uint32_t));
                if (e != hipSuccess) st = fail(CUDDLE_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e));
                if (st == CUDDLE_OK) {
                    e = hipMemcpy(raw, d->rabitq_codes, N * W32 * sizeof(uint32_t), hipMemcpyHostToDevice);
                    if (e != hipSuccess) st = fail(CUDDLE_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
                }
                src = raw;
            }
            if (st == CUDDLE_OK)
                hipLaunchKernelGGL(relayout_rabitq_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, src, rc,
                                   vec_off, d_pos_leaf, N, W32);
            // the same codes once more, vector-major (rows of whole 16-byte pieces): the exact estimates of the few
            // candidates the bounds leave undecided read one vector's code as one contiguous piece
            const uint32_t vmw = (W32 + 3) & ~3u;
            if (st == CUDDLE_OK) st = dev_alloc(ix, (size_t)N * vmw, &rvm);
            if (st == CUDDLE_OK) {
                const uint64_t total = N * (uint64_t)vmw, per = (1ull << 23) * 256;
                for (uint64_t t0 = 0; t0 < total; t0 += per)
                    hipLaunchKernelGGL(pad_rabitq_rows_kernel, dim3((unsigned)((std::min(per, total - t0) + 255) / 256)), dim3(256), 0, 0,
                                       src, rvm, N, W32, vmw, t0);
            }
            (void)hipDeviceSynchronize();
            if (raw) (void)hipFree(raw);
        }
        if (st == CUDDLE_OK) { v.rot_t = rot; v.rot = rot_tr; v.rq_norm = norm; v.rq_ipoo = ipoo; v.rq_codes = rc; v.rq_codes_vm = rvm; v.rq_vm_words = (W32 + 3) & ~3u; }
    }
    if (d_pos_leaf) (void)hipFree(d_pos_leaf);
    if (st != CUDDLE_OK) return st;
    HIP_TRY(hipMalloc((void **)&ix->d_stats, sizeof(DevStats)));
    HIP_TRY(hipMemset(ix->d_stats, 0, sizeof(DevStats)));
    ix->half_ok.assign(ix->n_levels, 1);
    ix->half_seen.assign(ix->n_levels, 0);
    HIP_TRY(hipMalloc((void **)&ix->d_half_fb, std::max<size_t>(ix->n_levels, 1) * 4));
    HIP_TRY(hipMemset(ix->d_half_fb, 0, std::max<size_t>(ix->n_levels, 1) * 4));
    HIP_TRY(hipHostMalloc((void **)&ix->h_half_fb, std::max<size_t>(ix->n_levels, 1) * 4, hipHostMallocDefault));
    memset(ix->h_half_fb, 0, std::max<size_t>(ix->n_levels, 1) * 4);
    HIP_TRY(hipEventCreateWithFlags(&ix->half_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ix->ws_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ix->fork_ev, hipEventDisableTiming));
    for (int l = 0; l < 2; ++l) {
        HIP_TRY(hipStreamCreateWithFlags(&ix->lane_stream[l], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ix->lane_ev[l], hipEventDisableTiming));
    }
    ST_TRY(scratch_for_device(device_id, &ix->scratch_ptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipGetLastError());
    // kernels that may need more than 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void *)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)selection_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)scan_kernel_for(v.dim, v.words), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)scan_all_hamming_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)select_rows_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)select_rows_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<256, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<256, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<512, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<512, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_select_reg_kernel<512, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)grouped_kernel_for(v.dim, v.metric), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)rabitq_query_prep_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)rabitq_query_prep_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)scan_select_rabitq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)scan_all_rabitq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)selection_sort_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // a query vector in LDS: more than 64 KiB from 16 385 dimensions on
    (void)hipFuncSetAttribute((const void *)rerank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_dist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)route_dist_roots_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // the leaf tables of the sharded selection steps: 16 bytes per leaf slot
    (void)hipFuncSetAttribute((const void *)shard_ties_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)shard_ties_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)shard_select_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)shard_select_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)rq_exchange_keys_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError(); // attribute failures are not fatal: launches that need the space report them
    guard.p = nullptr;
    *out = ix;
    return CUDDLE_OK;
}

extern "C" void cuddle_index_destroy(cuddle_index *idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    delete idx;
}

extern "C" uint32_t cuddle_index_dim(const cuddle_index *idx) { return idx ? idx->dev.dim : 0; }
extern "C" uint32_t cuddle_index_levels(const cuddle_index *idx) { return idx ? idx->n_levels : 0; }
extern "C" uint32_t cuddle_index_max_children(const cuddle_index *idx) { return idx ? idx->max_children : 0; }
extern "C" uint32_t cuddle_index_max_leaf_size(const cuddle_index *idx) { return idx ? idx->max_leaf : 0; }
extern "C" uint64_t cuddle_index_device_bytes(const cuddle_index *idx) {
    return idx ? idx->index_bytes + idx->ws.bytes + idx->io.bytes : 0;
}

// ---------------------------------------------------------------------------------------------
// planning the work space
// ---------------------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static bool no_half_env() { // tuning aid: split-bf16 filter on every level
    static const bool v = getenv("CUDDLE_ROUTE_SPLIT_FILTER") != nullptr;
    return v;
}

// Extra candidates the MFMA routing filter keeps beyond `probes` (they absorb the rounding error of
// the norm-expansion distances; too few only costs exact-path fallbacks, never correctness).
static uint32_t route_margin(uint32_t probes) {
    if (const char *e = getenv("CUDDLE_ROUTE_MARGIN")) return (uint32_t)std::max(0, atoi(e));
    return std::max<uint32_t>(24, probes / 8);
}
// ... with the single-pass half-precision filter, whose error bound is about 5 (960-d) to 14 (128-d) times the split one's
static uint32_t route_margin_half(uint32_t probes) {
    if (const char *e = getenv("CUDDLE_ROUTE_MARGIN_HALF")) return (uint32_t)std::max(0, atoi(e));
    return std::max<uint32_t>(32, probes / 6);
}

// CUDDLE_RABITQ_GROUPED = 0 / 1 forces the choice (A/B runs, tests); 2 = grouped with every bound checked against the
// per-query kernel's own (the search fails if one differs).
static int rq_grouped_env() {
    const char *e = getenv("CUDDLE_RABITQ_GROUPED"); // (read at every call: a test sets it for one search)
    return e ? atoi(e) : -1;
}
static bool rq_grouped_wanted(uint32_t nq, uint32_t dim, uint32_t lstride) {
    const int e = rq_grouped_env();
    if (e >= 0) return e != 0;
    // Off unless asked for. Measured on 10M x 768-d, 4096 queries x 100 leaves (4.2 queries per leaf): the bounds of all
    // pairs take 1.14 ms leaf by leaf against the 0.86 ms they take inside the per-query kernel - expanding a leaf's
    // bits is shared by too few queries, and what is left per pair (turning the integer into two bounds) costs as much
    // as the popcounts did. It pays where many queries share a leaf (small indexes, large batches).
    (void)nq;
    return false;
}

using LeafKernel8 = void (*)(IndexDev, const uint32_t *, const uint4 *, const uint32_t *, const uint2 *, const uint64_t *, uint8_t *, uint32_t);
using LeafKernel16 = void (*)(IndexDev, const uint32_t *, const uint4 *, const uint32_t *, const uint2 *, const uint64_t *, uint16_t *, uint32_t);
// The leaf-by-leaf Hamming kernel for a code of `words` u64 (one instantiation per width scan_kernel_for() knows); null otherwise.
static LeafKernel8 leaf_kernel8_for(uint32_t words) {
    switch (words) {
    case 1: return leaf_hamming_kernel<uint8_t, 0, true>;
    case 2: return leaf_hamming_kernel<uint8_t, 1, false>;
    case 3: return leaf_hamming_kernel<uint8_t, 1, true>;
    case 4: return leaf_hamming_kernel<uint8_t, 2, false>;
    default: return nullptr;
    }
}
static LeafKernel16 leaf_kernel16_for(uint32_t words) {
    switch (words) {
    case 4: return leaf_hamming_kernel<uint16_t, 2, false>;
    case 8: return leaf_hamming_kernel<uint16_t, 4, false>;
    case 12: return leaf_hamming_kernel<uint16_t, 6, false>;
    case 15: return leaf_hamming_kernel<uint16_t, 7, true>;
    case 16: return leaf_hamming_kernel<uint16_t, 8, false>;
    case 24: return leaf_hamming_kernel<uint16_t, 12, false>;
    default: return nullptr;
    }
}

// Leaf by leaf when the chunk's queries reach a leaf several times over: the leaf's codes are then read once per
// 64 queries instead of once per query, and the distances cost two bytes per pair to hand over. The number of
// (query, leaf) pairs is bounded by the leaf-list capacity; a chunk that cannot fill every leaf twice scans per query.
// CUDDLE_LEAF_GROUPED = 0 / 1 forces the choice (A/B runs, tests).
static bool leaf_grouped_wanted(const cuddle_index *ix, uint32_t nq, uint32_t lstride) {
    const IndexDev &v = ix->dev;
    if (!(v.dim < 255 ? (bool)leaf_kernel8_for(v.words) : (bool)leaf_kernel16_for(v.words))) return false;
    const char *e = getenv("CUDDLE_LEAF_GROUPED"); // (read at every call: a test sets it for one search)
    if (e) return atoi(e) != 0;
    // Measured (step time per query and leaf by leaf): 1000 queries x 232 of 9852 leaves of 960-d codes 0.68 -> 0.51 ms;
    // 1000 x 200 of 97 336 such leaves 0.92 -> 0.85 ms; 128-d codes lose (2500 x 150 of 9849 leaves 1.39 -> 1.48 ms,
    // 4096 x 100 of 125 000 leaves 0.56 -> 0.62 ms): a 16-byte code is cheaper to read again than to hand its distance
    // over, and their searches spend their time selecting, not reading. Codes of 512 bits and more, then.
    return v.words >= 8 && (uint64_t)nq * lstride >= 2ull * std::max<uint32_t>(ix->n_leaves, 1);
}

// Lays the per-chunk arrays of `pl` (strides and qc already set) out from `b`; returns the bytes they take.
// b == null: sizing only.
static size_t carve_plan(const cuddle_index *ix, Plan &pl, char *b) {
    const IndexDev &v = ix->dev;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t qc = pl.qc;
    const size_t dist_bytes = v.code_kind == CUDDLE_CODE_BINARY ? (v.dim < 255 ? 1 : 2) : 8; // u8 / u16 Hamming; f32 estimates + f32 lower bounds
    size_t o_qcodes = carve(qc * std::max<uint32_t>(v.words, v.dim + 2) * 8); // binary: W u64; RaBitQ: dim+2 floats
    size_t o_f0 = carve(qc * pl.fstride * 4), o_f1 = carve(qc * pl.fstride * 4);
    size_t o_fc0 = carve(qc * 4), o_fc1 = carve(qc * 4);
    size_t o_fd = carve(qc * pl.fstride * 4);
    size_t o_lv = carve(qc * pl.lstride * 4), o_lc = carve(qc * 4);
    size_t o_gd = carve(qc * (size_t)pl.cstride * dist_bytes);
    size_t o_sp = carve(qc * pl.sstride * 4), o_sk = carve(qc * pl.sstride * 4), o_sd = carve(qc * pl.sstride * 4);
    size_t o_sc = carve(qc * 4), o_cc = carve(qc * 4);
    size_t o_qcent = carve(qc * split_kp(v.dim) * 4), o_qsq = carve(qc * 4), o_qinv = carve(qc * 4);
    size_t o_rcp = carve(qc * pl.kc_cap * 4), o_rcd = carve(qc * pl.kc_cap * 4), o_rcm = carve(qc * 4), o_rfl = carve(qc * 4), o_rcc = carve(qc * 4);
    const size_t gw = (size_t)ix->max_parent_width + 1;
    size_t o_pi = carve(qc * pl.probes * 4), o_pp = carve(qc * pl.probes * 4), o_ps = carve(qc * pl.probes * 4), o_pl = carve(qc * pl.probes * 4);
    size_t o_sf = carve(qc * 4), o_gc = carve(gw * 4), o_gp = carve(gw * 4), o_gi = carve(gw * 4), o_gf = carve(gw * 4);
    size_t o_git = carve((qc * pl.probes / kGroupPairs + gw + 1) * 16);
    size_t o_thr = carve(qc * 8), o_sord = carve(qc * pl.sstride * 8), o_qn = carve(qc * 4);
    size_t o_lfc = carve(qc * ((size_t)ix->n_levels + 1) * 4);
    size_t o_ltb = carve(qc * 4 * ((size_t)pl.lstride + 4) * 4);
    size_t o_sg0 = carve(qc * seg_stride(pl.probes) * 4), o_sg1 = carve(qc * seg_stride(pl.probes) * 4);
    const bool rq = v.code_kind == CUDDLE_CODE_RABITQ;
    size_t o_rqr = carve(rq ? qc * (size_t)rq_plane_stride(v.words32) * 4 : 0);
    // the grouped bounds: sized for every (query, leaf-list slot) pair and every node (pl.rq_grouped decides their use)
    const bool rg = (rq && pl.rq_grouped) || pl.leaf_grouped;
    const size_t npair = rg ? qc * pl.lstride : 0, nnode = rg ? (size_t)v.n_nodes + 1 : 0;
    size_t o_ri8 = carve(rq && pl.rq_grouped ? qc * (size_t)rq_i8_stride(v.dim) : 0);
    size_t o_rpi = carve(npair * 8), o_rpn = carve(npair * 4), o_rps = carve(npair * 4), o_rpo = carve(npair * 8);
    size_t o_rc = carve((nnode + (rg ? 2 * kWorkBins : 0)) * 4), o_rpf = carve(nnode * 4), o_rio = carve(nnode * 4); // (+ the items' work classes)
    size_t o_rbs = carve(rg ? 2 * ((nnode + kPairScanChunk - 1) / kPairScanChunk + 1) * 4 : 0);
    size_t o_rit = carve(rg ? (npair / kGroupPairs + nnode + 1) * 16 : 0);
    size_t o_rod = carve(pl.leaf_grouped ? (npair / kGroupPairs + nnode + 1) * 4 : 0);
    size_t o_lk = carve(qc * 4), o_rro = carve(qc * 4);
    if (!b) return off;
    pl.qcodes = (uint64_t *)(b + o_qcodes);
    pl.frontier[0] = (uint32_t *)(b + o_f0); pl.frontier[1] = (uint32_t *)(b + o_f1);
    pl.fcount[0] = (uint32_t *)(b + o_fc0); pl.fcount[1] = (uint32_t *)(b + o_fc1);
    pl.fdist = (float *)(b + o_fd);
    pl.leaves = (uint32_t *)(b + o_lv); pl.lcount = (uint32_t *)(b + o_lc);
    pl.gdist = (uint16_t *)(b + o_gd);
    pl.sel_pos = (uint32_t *)(b + o_sp); pl.sel_key = (uint32_t *)(b + o_sk); pl.sel_dist = (float *)(b + o_sd);
    pl.sel_count = (uint32_t *)(b + o_sc); pl.cand_count = (uint32_t *)(b + o_cc);
    pl.qcent = (uint16_t *)(b + o_qcent); pl.qsq = (float *)(b + o_qsq); pl.qinv = (float *)(b + o_qinv);
    pl.rc_pos = (uint32_t *)(b + o_rcp); pl.rc_dist = (float *)(b + o_rcd); pl.rc_max = (float *)(b + o_rcm);
    pl.rflag = (uint32_t *)(b + o_rfl); pl.rc_cnt = (uint32_t *)(b + o_rcc);
    pl.pair_info = (uint32_t *)(b + o_pi); pl.pair_parent = (uint32_t *)(b + o_pp); pl.pair_sorted = (uint32_t *)(b + o_ps); pl.pair_slot = (uint32_t *)(b + o_pl);
    pl.small_flag = (uint32_t *)(b + o_sf);
    pl.grp_cnt = (uint32_t *)(b + o_gc); pl.grp_poff = (uint32_t *)(b + o_gp); pl.grp_ioff = (uint32_t *)(b + o_gi);
    pl.grp_fill = (uint32_t *)(b + o_gf);
    pl.grp_item = (uint32_t *)(b + o_git);
    pl.thr = (uint32_t *)(b + o_thr);
    pl.sel_ord = (uint64_t *)(b + o_sord);
    pl.qnorm = (float *)(b + o_qn);
    pl.level_fc = (uint32_t *)(b + o_lfc);
    pl.leaf_tables = (uint32_t *)(b + o_ltb);
    pl.seg[0] = (uint32_t *)(b + o_sg0); pl.seg[1] = (uint32_t *)(b + o_sg1);
    pl.rq_rows = (uint32_t *)(b + o_rqr);
    pl.rq_i8 = (int8_t *)(b + o_ri8);
    pl.rq_pinfo = (uint32_t *)(b + o_rpi); pl.rq_pnode = (uint32_t *)(b + o_rpn); pl.rq_pslot = (uint32_t *)(b + o_rps);
    pl.rq_psorted = (uint32_t *)(b + o_rpo);
    pl.rq_cnt = (uint32_t *)(b + o_rc); pl.rq_poff = (uint32_t *)(b + o_rpf); pl.rq_ioff = (uint32_t *)(b + o_rio);
    pl.rq_bsum = (uint32_t *)(b + o_rbs); pl.rq_items = (uint32_t *)(b + o_rit);
    pl.rq_order = (uint32_t *)(b + o_rod);
    pl.loc_key = (uint32_t *)(b + o_lk); pl.rr_order = (uint32_t *)(b + o_rro);
    return off;
}

// Sums the per-query counts that the last single-chunk search left in the work space into the device counters
// (on that search's stream), if that has not happened yet.
static void flush_lazy_stats(cuddle_index *ix) {
    if (!ix->lazy_pending) return;
    hipLaunchKernelGGL(stats_accum_kernel, dim3(std::min<uint32_t>((ix->lazy_nq + 255) / 256, 8)), dim3(256), 0, ix->last_stream,
                       ix->lazy_lcount, ix->lazy_cand, ix->lazy_sel, ix->lazy_nq, ix->d_stats);
    ix->lazy_pending = false;
}

// lstride_min: lower bound for the leaf-list capacity (stage entry points pass their own lists).
// fixed_base != null: the work space is that device buffer of fixed_cap bytes (nothing is allocated).
// whole_index: the selection is made over this index alone, so a selection size beyond the candidates a query can
// reach means "all of them" and is cut to that bound (a shard of a sharded search keeps the size the ranks agreed on).
static cuddle_status make_plan(cuddle_index *ix, uint32_t nq, uint32_t k, uint32_t probes, uint64_t rerank_k64,
                               uint32_t lstride_min, Plan &pl, void *fixed_base = nullptr, size_t fixed_cap = 0,
                               bool whole_index = true) {
    const IndexDev &v = ix->dev;
    if (probes > kSortCap) {
        // probes beyond every level's width behave as "keep / sort everything"; that is fine as
        // long as no level is wider than the sort capacity.
        uint32_t widest = 0;
        for (uint32_t w : ix->level_width) widest = std::max(widest, w);
        if (widest > kSortCap)
            return fail(CUDDLE_ERR_UNSUPPORTED, "probes_per_level %u > %u with a level of %u nodes", probes, kSortCap, widest);
        probes = kSortCap;
    }
    pl.qc = std::min(nq, kMaxChunk);
    pl.k = k;
    pl.probes = probes;
    // frontier bound: level 0 = roots; level l+1 <= min(width, min(probes, bound_l) * max_children)
    uint64_t fb = v.n_roots, fmax = v.n_roots, lsum = 0;
    for (uint32_t l = 0; l < ix->n_levels; ++l) {
        uint64_t width = ix->level_width[l];
        fb = std::min<uint64_t>(fb, width);
        fmax = std::max(fmax, fb);
        uint64_t sel = std::min<uint64_t>(fb, probes);
        lsum += std::min<uint64_t>(sel, ix->level_leaves[l]); // a level contributes at most its own leaves
        fb = sel * std::max<uint32_t>(ix->max_children, 1);
    }
    pl.fstride = (uint32_t)align_up(fmax, 32); // rows of the distance matrix start on 128-byte lines: the contraction's 128-byte stores stay whole
    pl.lstride = (uint32_t)align_up(std::max<uint64_t>(std::min<uint64_t>(lsum, ix->n_leaves), std::max<uint32_t>(lstride_min, 1)), 4);
    uint64_t cbound = std::min<uint64_t>((uint64_t)pl.lstride * ix->max_leaf, v.n_vectors);
    if (cbound >= kMaxCandidates || pl.lstride >= (1u << 20))
        return fail(CUDDLE_ERR_UNSUPPORTED, "a query may reach %llu candidates in %u leaves; at most %u candidates per query",
                    (unsigned long long)cbound, pl.lstride, kMaxCandidates - 1);
    if (whole_index) rerank_k64 = std::min<uint64_t>(rerank_k64, std::max<uint64_t>(cbound, 1));
    if (rerank_k64 > kMaxRerankK)
        return fail(CUDDLE_ERR_UNSUPPORTED, "a selection of %llu candidates per query (k*rerank_factor, or all a query can reach) > %u",
                    (unsigned long long)rerank_k64, kMaxRerankK);
    pl.rerank_k = (uint32_t)rerank_k64;
    pl.cstride = (uint32_t)align_up(std::max<uint64_t>(cbound, 8), 128); // spill rows never share a cache line
    pl.sstride = (uint32_t)align_up(std::max<uint64_t>(std::min<uint64_t>(pl.rerank_k, cbound), 4), 4);
    pl.tcap = (uint32_t)align_up((size_t)pl.lstride + pl.cstride / 64 + 1, 4);
    // small per-query scans (few candidates): smaller blocks, more of them per CU, so that the
    // latency-bound prologue/selection phases of one query overlap the streaming of others
    // (two-word codes are built for 8 waves per SIMD: whole-CU occupancy with 512-thread blocks)
    pl.leaf_grouped = v.code_kind == CUDDLE_CODE_BINARY && leaf_grouped_wanted(ix, std::min(nq, kMaxChunk), pl.lstride);
    pl.scan_threads = (pl.cstride <= 16384 && v.words != 2) ? 256 : kScanThreads;
    if (v.code_kind == CUDDLE_CODE_RABITQ) pl.scan_threads = 1024; // two blocks of 16 waves per CU: the LDS holds no more blocks, the registers this many waves
    if (const char *e = getenv("CUDDLE_SCAN_THREADS")) {  // tuning aid; the kernels are bounded at 512 threads
        const int t = atoi(e);
        if (t >= 64 && t <= (int)kScanThreads && t % 64 == 0) pl.scan_threads = (uint32_t)t;
    }
    if (v.code_kind == CUDDLE_CODE_BINARY) {
        const uint32_t esz = v.dim < 255 ? 1 : 2; // one byte per distance when dim < 255
        pl.dist_cap = std::min<uint32_t>(pl.cstride, kLdsDistCap * 2 / esz);
        // cstride is the worst case (every reached leaf as large as the largest); a distance array of that size can
        // halve the blocks a CU holds (SIFT1M shape: 57 KB per block, 2 blocks of 8 waves where the registers allow
        // 4; 0.379 -> 0.274 ms per 10 000 queries with 4). So: the most blocks per CU whose share of the LDS still
        // holds half as many candidates again as a query reaches on average; a query beyond that keeps its distances
        // in its row of the spill buffer, as one beyond the old size did.
        const size_t fixed = scan_lds_bytes(v.dim, v.words, pl.lstride, pl.tcap, 0, v.words == 2);
        // (on average: a leaf of average size for every probe of the last level; lstride also counts what every
        // level above could add)
        const uint64_t typical = (uint64_t)std::min(pl.lstride, pl.probes) * v.n_vectors / std::max<uint32_t>(ix->n_leaves, 1) * 3 / 2;
        for (uint32_t blocks = 2048 / pl.scan_threads; blocks >= 1; --blocks) {
            const size_t share = 160 * 1024 / blocks - 512;
            if (share <= fixed) continue;
            const uint32_t fit = (uint32_t)((share - fixed) / esz) & ~127u;
            if (fit >= std::min<uint64_t>(pl.dist_cap, typical) || blocks == 1) {
                pl.dist_cap = std::min(pl.dist_cap, std::max(fit, 128u));
                break;
            }
        }
        if (const char *e = getenv("CUDDLE_LDS_DIST_BYTES")) // tuning aid
            pl.dist_cap = std::max<uint32_t>(128, std::min<uint32_t>(std::min<uint32_t>(pl.cstride, kLdsDistCap * 2 / esz), (uint32_t)atoi(e) / esz));
        pl.scan_lds = scan_lds_bytes(v.dim, v.words, pl.lstride, pl.tcap, (pl.dist_cap * esz + 1) / 2, v.words == 2);
    } else {
        pl.dist_cap = std::min(pl.cstride, kLdsDistCap / 2); // f32 estimates
        pl.scan_lds = rabitq_scan_lds_bytes(v.dim, pl.lstride, pl.tcap, pl.dist_cap);
    }
    if (pl.scan_lds > 150 * 1024) return fail(CUDDLE_ERR_UNSUPPORTED, "leaf list of %u slots does not fit the scan kernel's LDS", pl.lstride);
    // RaBitQ bounds leaf by leaf on the matrix cores: pays once a leaf is shared by several queries of the chunk, i.e. for
    // batches (the per-query kernel computes its own bounds otherwise); dimensions the kernel has an instantiation for
    pl.rq_grouped = v.code_kind == CUDDLE_CODE_RABITQ && rq_grouped_wanted(nq, v.dim, pl.lstride);
    pl.kc_split = (uint32_t)align_up((size_t)pl.probes + route_margin(pl.probes), 4); // every candidate costs an exact distance
    pl.kc_half = (uint32_t)align_up((size_t)pl.probes + route_margin_half(pl.probes), 4);
    pl.kc_cap = std::max(pl.kc_split, pl.kc_half);
    // Chunk size: the per-query rows (above all the spill row of phase-1 distances) times the chunk must fit
    // the budget - the caller's fixed buffer, or kWorkspaceBudget of device memory for the handle's own.
    const size_t budget = fixed_base ? fixed_cap : kWorkspaceBudget;
    while (pl.qc > 1 && carve_plan(ix, pl, nullptr) > budget) pl.qc = (pl.qc + 1) / 2;
    pl.bytes = carve_plan(ix, pl, nullptr);
    if (fixed_base) {
        if (pl.bytes > fixed_cap)
            return fail(CUDDLE_ERR_UNSUPPORTED, "one query of this search needs %zu bytes of work space; the fixed buffer holds %zu",
                        pl.bytes, fixed_cap);
        carve_plan(ix, pl, static_cast<char *>(fixed_base));
    } else {
        // growing the work space frees the old one: the per-query counts a search left there for cuddle_last_stats
        // are summed first (a search re-plans after enter_stream() has dropped them; cuddle_index_reserve has not)
        if (pl.bytes > ix->ws.bytes) flush_lazy_stats(ix);
        ST_TRY(ix->ws.ensure(pl.bytes));
        carve_plan(ix, pl, static_cast<char *>(ix->ws.p));
    }
    return CUDDLE_OK;
}

static uint32_t plan_leaf_stride(const cuddle_index *ix, uint32_t probes);

static cuddle_status rerank_k_of(uint32_t k, uint32_t rf, uint32_t flags, uint64_t &out) {
    if (k == 0) return fail(CUDDLE_ERR_BAD_ARG, "k must be >= 1 (the reference underflows k-1, hierarchical.rs:651)");
    if (rf == 0) return fail(CUDDLE_ERR_BAD_ARG, "rerank_factor must be >= 1");
    uint64_t r = (uint64_t)k * rf;
    if (!(flags & CUDDLE_SEARCH_STATS_VARIANT)) r = std::min<uint64_t>(r, 10000); // hierarchical.rs:447
    out = r;
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// stage launchers (device pointers, asynchronous on `s`)
// ---------------------------------------------------------------------------------------------
// A handle has ONE work space: a call on another stream than the previous call's is ordered behind that stream
// (an event recorded there now, awaited by `s`: no host wait), so that it cannot overwrite the work space of a batch
// that is still running. A stream that is recording a graph is left alone: whoever replays the graph orders the replays.
static cuddle_status enter_stream(cuddle_index *ix, hipStream_t s) {
    ix->lazy_pending = false; // the call about to run re-plans the work space (and clears the counters anyway)
    if (stream_capturing(s)) return CUDDLE_OK;
    if (ix->stream_used && ix->last_stream != s) order_after(ix->last_stream, s, ix->ws_ev);
    ix->last_stream = s;
    ix->stream_used = true;
    return CUDDLE_OK;
}

// Records an event; `stage` names the work enqueued between it and the next event (-1: none).
static void mark(cuddle_index *ix, hipStream_t s, int stage) {
    if (!ix->timed) return;
    if (ix->events_used == ix->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ix->events.push_back(e);
        ix->event_stage.push_back(-1);
    }
    ix->event_stage[ix->events_used] = stage;
    (void)hipEventRecord(ix->events[ix->events_used++], s);
}


static uint32_t host_next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// route_init != null: the launch also sets up the routing state of the chunk (what route_init_kernel does; *route_init
// tells launch_route that it has been done), clearing the handle's counters when zero_stats is set. Binary codes only:
// *route_init stays false otherwise.
static cuddle_status launch_quantize(cuddle_index *ix, const Plan &pl, const float *dq, uint32_t nq, hipStream_t s,
                                     bool *route_init = nullptr, uint32_t flags = 0, bool zero_stats = false) {
    const IndexDev &v = ix->dev;
    if (route_init) *route_init = false;
    if (v.code_kind == CUDDLE_CODE_BINARY) {
        uint64_t total = (uint64_t)nq * v.words;
        const uint32_t qblocks = (uint32_t)((total + 3) / 4), rblocks = (v.n_roots + 255) / 256;
        if (route_init && (uint64_t)qblocks + (uint64_t)rblocks * nq < (1u << 30)) {
            hipLaunchKernelGGL(quantize_init_kernel, dim3(qblocks + rblocks * nq), dim3(256), 0, s, dq, (uint64_t)nq, v.dim, v.words,
                               ix->threshold, pl.qcodes, qblocks, rblocks, v.roots, v.n_roots, pl.frontier[0], pl.fstride, pl.fcount[0],
                               pl.lcount, nq, (flags & CUDDLE_SEARCH_STATS_VARIANT) ? 1u : 0u,
                               zero_stats ? reinterpret_cast<uint32_t *>(ix->d_stats) : nullptr, (uint32_t)(sizeof(DevStats) / 4));
            *route_init = true;
        } else {
        hipLaunchKernelGGL(binary_quantize_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, dq, (uint64_t)nq,
                           v.dim, v.words, ix->threshold, pl.qcodes);
        }
    } else {
        float *rq = reinterpret_cast<float *>(pl.qcodes), *qn = rq + (size_t)pl.qc * v.dim;
        const dim3 pgrid((nq + kPrepQ - 1) / kPrepQ, (v.dim + kPrepRows - 1) / kPrepRows);
        if (pgrid.x * pgrid.y <= 256) // fewer blocks than CUs: each thread's chain of loads is the whole cost
            hipLaunchKernelGGL(rabitq_query_prep_kernel<64>, pgrid, dim3(kPrepRows), kPrepQ * v.dim * sizeof(float), s, dq, v.rot, nq, v.dim, rq, qn);
        else
            hipLaunchKernelGGL(rabitq_query_prep_kernel<32>, pgrid, dim3(kPrepRows), kPrepQ * v.dim * sizeof(float), s, dq, v.rot, nq, v.dim, rq, qn);
        hipLaunchKernelGGL(rabitq_query_planes_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, rq, qn, nq, v.dim, v.words32, pl.rq_rows,
                           pl.rq_grouped ? pl.rq_i8 : (int8_t *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

// Cosine: the query-side norms, once per batch instead of once per (query, stored vector).
static void launch_query_norms(cuddle_index *ix, const Plan &pl, const float *dq, uint32_t nq, hipStream_t s) {
    if (ix->dev.metric != CUDDLE_METRIC_COSINE) return;
    hipLaunchKernelGGL(query_norm_kernel, dim3((nq * 4 + 255) / 256), dim3(256), 0, s, dq, nq, ix->dev.dim, pl.qnorm);
}

// Prefilter of the approximate distances in pl.fdist: keys in registers, as few per thread as the bound
// on the frontier width allows (unused key slots still cost their predicated instructions in every pass).
// ncount / skip / cand_cnt: per-query frontiers (see route_prefilter_reg_kernel).
static void launch_prefilter(const Plan &pl, uint32_t nq, uint32_t n, uint32_t kc, const uint32_t *ncount, const uint32_t *skip,
                             uint32_t *cand_cnt, hipStream_t s) {
#define CUDDLE_PREFILTER(T, K)                                                                                                \
    hipLaunchKernelGGL((route_prefilter_reg_kernel<T, K>), dim3(nq), dim3(T), 0, s, pl.fdist, pl.fstride, n, kc, pl.rc_pos,    \
                       pl.kc_cap, pl.rc_max, ncount, skip, cand_cnt)
    if (n <= 128 * 8) CUDDLE_PREFILTER(128, 8);
    else if (n <= 256 * 8) CUDDLE_PREFILTER(256, 8);
    else if (n <= 256 * 16) CUDDLE_PREFILTER(256, 16);
    else if (n <= 256 * 24) CUDDLE_PREFILTER(256, 24);
    else if (n <= 512 * 16) CUDDLE_PREFILTER(512, 16);
    else if (n <= 512 * 20) CUDDLE_PREFILTER(512, 20);
    else if (n <= 512 * 24) CUDDLE_PREFILTER(512, 24);
    else CUDDLE_PREFILTER(512, 32);
#undef CUDDLE_PREFILTER
}

// Approximate distances of every query of the chunk to every node of a level that has contraction
// operands: pl.fdist[q][0..g.n) and pl.qsq[q].
// half: from the single-pass half-precision operands (g.table16 must exist) instead of the split ones.
static void launch_level_gemm(cuddle_index *ix, const Plan &pl, const LevelGemm &g, const float *dq, uint32_t nq, hipStream_t s,
                              bool half = false) {
    const IndexDev &v = ix->dev;
    const uint32_t n = g.n;
    const uint32_t kp = split_kp(v.dim);
    const uint32_t gy = (nq + kGemmBM - 1) / kGemmBM, gx = (n + 63) / 64;
    // XCD row groups: as few as keep an XCD's share of A within half of its 4 MiB L2
    uint32_t rg = 1;
    while (rg < 8 && (uint64_t)nq * kp * (half ? 2 : 4) / rg > (2u << 20) && gy >= 2 * rg) rg *= 2;
    const uint32_t rows_per = (gy + rg - 1) / rg, pgroups = 8 / rg;
    if (half) {
        hipLaunchKernelGGL(center_queries16_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, dq, g.mu, nq, v.dim, kp,
                           reinterpret_cast<_Float16 *>(pl.qcent), pl.qsq, pl.qinv);
#define CUDDLE_ROUTE_GEMM16(NT, BK)                                                                                          \
    hipLaunchKernelGGL((route_gemm16_kernel<NT, BK>), dim3(8 * rows_per * ((gxh + pgroups - 1) / pgroups)), dim3(256),         \
                       (route_gemm16_lds_bytes<NT, BK>()), s, (const u32x4 *)pl.qcent, pl.qsq, pl.qinv, (const u32x4 *)g.table16, \
                       g.sqn, g.inv16, nq, n, kp, v.metric, pl.fdist, pl.fstride, gxh, gy, rg)
        // measured on 1000 x 8682 x 960: K steps of 128 the same within 1 %, 128-wide column tiles 2 % slower
        const uint32_t gxh = gx;
        CUDDLE_ROUTE_GEMM16(1, 64);
#undef CUDDLE_ROUTE_GEMM16
        return;
    }
    hipLaunchKernelGGL(center_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, dq, g.mu, nq, v.dim, kp, pl.qcent, pl.qsq);
#define CUDDLE_ROUTE_GEMM(NT, BK)                                                                                            \
    hipLaunchKernelGGL((route_gemm_kernel<NT, BK>), dim3(8 * rows_per * ((gx + pgroups - 1) / pgroups)), dim3(256),          \
                       (route_gemm_lds_bytes<NT, BK>()), s, (const u32x4 *)pl.qcent, pl.qsq, (const u32x4 *)g.table, g.sqn,  \
                       nq, n, kp, v.metric, pl.fdist, pl.fstride, gx, gy, rg)
    // Column tiles of 64 (NT = 1): the wider tile's better reuse never made up for its fewer, longer
    // blocks on the measured shapes. K step: 64 halves the barriers of a long contraction; short ones
    // are mostly epilogue and do better with the smaller LDS footprint (more blocks per CU) of 32.
    if (kp >= 512) CUDDLE_ROUTE_GEMM(1, 64); else CUDDLE_ROUTE_GEMM(1, 32);
#undef CUDDLE_ROUTE_GEMM
}

// HalfFeedback. route_finish_kernel counts, per level, the queries whose half-precision selection could not be proven
// exact. A snapshot of the counts travels to pinned host memory behind the first few routing calls and every 16th one
// after; a later call that finds the copy complete judges the levels: more than 2 % of the queries routed in between
// (and at least 4) failed -> the level uses the split filter from then on. Nothing waits for anything.
static void half_feedback_poll(cuddle_index *ix) {
    if (!ix->half_pending || hipEventQuery(ix->half_ev) != hipSuccess) return;
    ix->half_pending = false;
    const uint64_t dq = ix->half_q_posted - ix->half_q_seen;
    for (uint32_t l = 0; l < ix->n_levels; ++l) {
        const uint32_t d = ix->h_half_fb[l] - ix->half_seen[l];
        if (d >= 4 && (uint64_t)d * 50 > dq) ix->half_ok[l] = 0;
        ix->half_seen[l] = ix->h_half_fb[l];
    }
    ix->half_q_seen = ix->half_q_posted;
}
static void half_feedback_post(cuddle_index *ix, uint32_t nq, hipStream_t s) {
    ix->half_q_total += nq;
    ix->half_calls++;
    if (ix->half_pending || !(ix->half_calls <= 4 || ix->half_calls % 16 == 0)) return;
    if (hipMemcpyAsync(ix->h_half_fb, ix->d_half_fb, (size_t)ix->n_levels * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return;
    if (hipEventRecord(ix->half_ev, s) != hipSuccess) return;
    ix->half_pending = true;
    ix->half_q_posted = ix->half_q_total;
}

// All tree levels (hierarchical.rs:349-423). Leaves end up in pl.leaves / pl.lcount.
// A level whose frontier is the same for every query (all earlier levels kept all their nodes)
// and wider than probes goes through the MFMA contraction + exact refinement; every other level
// through the exact kernels. Both give the same selection.
// feedback: the call may enqueue a copy and an event for HalfFeedback (not for callers that must enqueue kernels only).
// zero_stats: the first kernel also clears the handle's device counters (the caller then needs no memset for them).
// init_done: the routing state has been set up already (launch_quantize did it in its own launch).
static cuddle_status launch_route(cuddle_index *ix, const Plan &pl, const float *dq, uint32_t nq, uint32_t flags,
                                  hipStream_t s, bool feedback = true, bool zero_stats = false, bool init_done = false) {
    const IndexDev &v = ix->dev;
    const uint32_t stats = (flags & CUDDLE_SEARCH_STATS_VARIANT) ? 1u : 0u;
    if (feedback && stream_capturing(s)) feedback = false; // event queries and copies to the host have no place in a graph
    if (feedback) half_feedback_poll(ix);
    bool used_half = false;
    auto half_on = [&](uint32_t l) { // the level's filter: single-pass half precision unless told or taught otherwise
        return ix->lvl[l].table16 && !(flags & CUDDLE_SEARCH_ROUTE_SPLIT_FILTER) && !no_half_env() && ix->half_ok[l];
    };
    // `shared`: every query's frontier at this level is the level's full node list, in
    // breadth-first order (true while all earlier levels kept all their nodes, search() only:
    // search_with_stats re-orders every level).
    bool shared = !stats;
    launch_query_norms(ix, pl, dq, nq, s);
    if (!init_done)
    hipLaunchKernelGGL(route_init_kernel, dim3((v.n_roots + 255) / 256, nq), dim3(256), 0, s, v.roots, v.n_roots,
                       pl.frontier[0], pl.fstride, pl.fcount[0], pl.lcount, nq, shared ? 0u : 1u,
                       zero_stats ? reinterpret_cast<uint32_t *>(ix->d_stats) : nullptr, (uint32_t)(sizeof(DevStats) / 4));
    int cur = 0;
    uint64_t fb = v.n_roots;
    bool grouped = false; // this level's distances come from the (query, parent) pairs of the level above
    bool have_seg = false; // ... and the grouped queries' frontiers exist as segment tables only (pl.seg[cur])
    static const bool no_filter = getenv("CUDDLE_GROUPED_EXACT") != nullptr; // tuning aid: exact distances for every pair
    // the filter paths end in route_finish_kernel, whose LDS holds the sorted candidates, the expansion scratch and the
    // segment table: with probes in the thousands and a level wider still that passes what a CU has, and the level
    // goes through the exact kernels instead
    auto finish_fits = [&](uint64_t width) {
        return route_finish_lds_bytes(host_next_pow2((uint32_t)std::min<uint64_t>(width, pl.kc_cap)), pl.probes) <= 160 * 1024;
    };
    for (uint32_t l = 0; l < ix->n_levels; ++l) {
        if (stats) // SearchStats::probes_per_level (hierarchical.rs:861): the frontier size of every query at this level
            HIP_TRY(hipMemcpyAsync(pl.level_fc + (size_t)l * pl.qc, pl.fcount[cur], (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
        fb = std::min<uint64_t>(fb, ix->level_width[l]);
        const uint32_t bound = (uint32_t)std::max<uint64_t>(fb, 1);
        const bool keep_all = !stats && bound <= pl.probes;
        // frontiers of up to scap nodes are sorted whole, larger ones are cut to probes first: scap
        // only has to hold probes, and a small one keeps the selection kernels' LDS footprint low
        const uint32_t scap = std::min(std::min(kSortCap, host_next_pow2(bound)), std::max(512u, host_next_pow2(pl.probes)));
        const LevelGemm &g = ix->lvl[l];
        const uint32_t *sfr = shared ? g.nodes : nullptr;
        const uint32_t sn = shared ? ix->level_width[l] : 0;
        const bool next_shared = shared && keep_all;
        const uint64_t fb_next = std::min<uint64_t>(fb, pl.probes) * std::max<uint32_t>(ix->max_children, 1);
        if (shared && keep_all && ix->level_leaves[l] == 0) {
            // every node of the level is kept and none is a leaf: nothing to compute or record
            fb = fb_next;
            continue;
        }
        const uint32_t *only = nullptr;
        bool grouped_filter = false, grouped_finish = false, grouped_half = false;
        uint32_t grouped_kc = 0;
        if (grouped) {
            // pairs recorded by the level above -> parent order -> one block per (parent, 8 queries)
            const uint32_t psh = pair_shift_of(ix->level_width[l - 1]); // counters per parent of the level above (PairOut::shift)
            const uint32_t wprev = ix->level_width[l - 1] << psh, total = nq * pl.probes;
            grouped_filter = !stats && g.n && bound <= 512 * 32 && !no_filter && finish_fits(bound);
            const uint32_t nb = (wprev + kPairScanChunk - 1) / kPairScanChunk; // pl.grp_fill: 2 * nb block sums / offsets
            const bool self_carry = nb <= 16; // few chunks: pair_scan_kernel sums the chunks before its own itself
            if (!self_carry) {
                hipLaunchKernelGGL(pair_block_sums_kernel, dim3(nb), dim3(256), 0, s, pl.grp_cnt, wprev, pl.grp_fill, nb);
                hipLaunchKernelGGL(pair_block_scan_kernel, dim3(1), dim3(1024), 0, s, pl.grp_fill, nb, wprev, pl.grp_poff, pl.grp_ioff);
            }
            hipLaunchKernelGGL(pair_scan_kernel, dim3(nb), dim3(256), 0, s, pl.grp_cnt, wprev, self_carry ? nullptr : pl.grp_fill, nb, pl.grp_poff, pl.grp_ioff,
                               reinterpret_cast<uint4 *>(pl.grp_item), grouped_filter ? ix->lvl[l - 1].child_rows : nullptr,
                               (uint32_t *)nullptr, (const uint32_t *)nullptr, psh);
            hipLaunchKernelGGL(pair_fill_kernel, dim3((total + 255) / 256), dim3(256), 0, s, pl.pair_info, pl.pair_parent, pl.pair_slot,
                               total, pl.grp_poff, pl.pair_sorted);
            const uint32_t items = (total + kGroupPairs - 1) / kGroupPairs + std::min(wprev, total);
            if (grouped_filter) {
                // approximate distances of all pairs on the matrix cores, then the same prefilter / exact
                // refinement / proof as on a shared level, per query; queries that are not grouped (small
                // frontiers) or fail the proof are flagged for the exact kernels below
                // half-precision operands where the contraction streams the level's table from HBM (it exists only for
                // such levels, see grouped_half_min_bytes): measured 170 -> 132 us on a 1M-node level of 128-d rows; where
                // the table stays on the die the split operands are as fast and need fewer exact distances
                grouped_half = half_on(l) && (uint64_t)g.n * split_kp(v.dim) * 4 > grouped_half_min_bytes();
                used_half |= grouped_half;
                const uint32_t kp = split_kp(v.dim), kc = std::min(bound, grouped_half ? pl.kc_half : pl.kc_split);
                if (grouped_half) {
                    hipLaunchKernelGGL(center_queries16_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, dq, g.mu, nq, v.dim, kp,
                                       reinterpret_cast<_Float16 *>(pl.qcent), pl.qsq, pl.qinv);
#define CUDDLE_GROUPED_GEMM16(KC)                                                                                            \
    hipLaunchKernelGGL(route_gemm16_grouped_kernel<KC>, dim3(items), dim3(256), (route_gemm16_grouped_lds_bytes<KC>()), s, v,   \
                       (const u32x4 *)pl.qcent, pl.qsq, pl.qinv, (const u32x4 *)g.table16, g.sqn, g.inv16, kp, wprev,           \
                       pl.grp_ioff, reinterpret_cast<const uint4 *>(pl.grp_item), pl.pair_sorted, pl.fdist, pl.fstride)
                    CUDDLE_GROUPED_GEMM16(64); // K chunk: 32 within 1 %, 128 slower
#undef CUDDLE_GROUPED_GEMM16
                } else {
                hipLaunchKernelGGL(center_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, dq, g.mu, nq, v.dim, kp, pl.qcent, pl.qsq);
#define CUDDLE_GROUPED_GEMM(KC)                                                                                              \
    hipLaunchKernelGGL(route_gemm_grouped_kernel<KC>, dim3(items), dim3(256), (route_gemm_grouped_lds_bytes<KC>()), s, v,       \
                       (const u32x4 *)pl.qcent, pl.qsq, (const u32x4 *)g.table, g.sqn, kp, wprev, pl.grp_ioff,                  \
                       reinterpret_cast<const uint4 *>(pl.grp_item), pl.pair_sorted, pl.fdist, pl.fstride)
                CUDDLE_GROUPED_GEMM(64); // K chunk: 32 and 64 measured alike, 128 slower (fewer blocks in flight)
#undef CUDDLE_GROUPED_GEMM
                }
                launch_prefilter(pl, nq, bound, kc, pl.fcount[cur], pl.small_flag, pl.rc_cnt, s);
                hipLaunchKernelGGL(route_refine_kernel, dim3((kc + 63) / 64, nq), dim3(256),
                                   (((size_t)v.dim + 3) & ~(size_t)3) * sizeof(float) + (have_seg ? seg_stride(pl.probes) * 4 : 0), s, v, dq,
                                   pl.frontier[cur], pl.rc_pos, pl.kc_cap, kc, pl.rc_dist, pl.qnorm, pl.fstride, pl.small_flag, pl.rc_cnt,
                                   have_seg ? pl.seg[cur] : nullptr, pl.probes);
                grouped_finish = true; // launched below, once this level's PairOut is known
                grouped_kc = kc;
                only = pl.rflag;
            } else {
                const size_t glds = 2 * (size_t)kGroupNQ * ((v.dim + 3) & ~3u) * sizeof(float);
                auto gk = grouped_kernel_for(v.dim, v.metric);
                hipLaunchKernelGGL(gk, dim3(items), dim3(256), glds, s, v, dq, ix->lvl[l - 1].nodes, wprev, pl.grp_ioff,
                                   reinterpret_cast<const uint4 *>(pl.grp_item), pl.pair_sorted, pl.fdist, pl.fstride, pl.qnorm);
                only = pl.small_flag; // queries with small frontiers: the exact kernel below, in their own arithmetic
            }
        }
        // Will the next level evaluate per-query frontiers wider than probes? Then this level's
        // expansion also records its (query, chosen node) pairs, grouped by node for that level.
        PairOut po;
        bool group_next = false, seg_next = false, materialise_pending = false;
        // (the selection kernel can take over the exact distances, and the frontiers, of the queries a filter path leaves)
        const bool fuse_ok = bound <= 512 * 32 && route_select_reg_lds_bytes(scap, v.dim) <= 150 * 1024 &&
                             (size_t)seg_stride(pl.probes) * 4 <= (size_t)scap * 8 + 4 * 256 * 4;
        const bool use_gemm = shared && !keep_all && !(flags & CUDDLE_SEARCH_ROUTE_EXACT_ONLY) && g.n && g.n == bound && finish_fits(g.n);
        uint32_t *zero_cnt = nullptr; // the next level's pair counters, cleared by this level's refinement kernel when it has one
        if (l + 1 < ix->n_levels && !next_shared && !(flags & CUDDLE_SEARCH_ROUTE_EXACT_ONLY) && pl.fstride <= (1u << kPairOffBits)) {
            // grouping has a fixed cost (pair sort, one block per 64 pairs): worth it when a query's
            // next frontier is expected to be wide, i.e. chosen nodes x mean fan-out of this level
            const uint64_t bound_next = std::min<uint64_t>(fb_next, ix->level_width[l + 1]);
            const uint64_t expect_next = std::min<uint64_t>(fb, pl.probes) * ix->level_width[l + 1] / std::max<uint32_t>(ix->level_width[l], 1);
            if ((stats || bound_next > std::max<uint32_t>(pl.probes, 100)) && (expect_next >= 256 || (flags & CUDDLE_SEARCH_ROUTE_GROUPED))) {
                group_next = true;
                po.info = pl.pair_info; po.parent = pl.pair_parent; po.slot = pl.pair_slot; po.cnt = pl.grp_cnt; po.small = pl.small_flag;
                po.pstride = pl.probes;
                po.shift = pair_shift_of(ix->level_width[l]);
                po.min_n = stats ? 0u : std::max<uint32_t>(pl.probes, 100);
                // the next level will go through the contraction (the condition of grouped_filter there): its kernels
                // need a frontier position's node for a few hundred positions per query, not the frontier itself
                seg_next = !stats && ix->lvl[l + 1].n && bound_next <= 512 * 32 && !no_filter && finish_fits(bound_next);
                if (seg_next) po.seg = pl.seg[cur ^ 1];
                if (use_gemm) zero_cnt = pl.grp_cnt;
                else HIP_TRY(hipMemsetAsync(pl.grp_cnt, 0, (((size_t)ix->level_width[l] << po.shift) + 1) * 4, s));
            }
        }
        if (grouped_finish) {
            hipLaunchKernelGGL(route_finish_kernel, dim3(nq), dim3(256), route_finish_lds_bytes(host_next_pow2(grouped_kc), pl.probes), s,
                               v, pl.frontier[cur], bound, pl.rc_pos, pl.rc_dist, pl.kc_cap, grouped_kc, pl.rc_max, pl.qsq, g.max_sqn,
                               pl.probes, pl.frontier[cur ^ 1], pl.fstride, pl.fcount[cur ^ 1], pl.leaves, pl.lstride, pl.lcount,
                               pl.rflag, &ix->d_stats->fallbacks, po, pl.fcount[cur], pl.small_flag, pl.rc_cnt,
                               have_seg ? pl.seg[cur] : nullptr, grouped_half ? 1u : 0u, grouped_half ? ix->d_half_fb + l : nullptr);
            materialise_pending = have_seg;
            if (have_seg && !fuse_ok) // queries that failed the proof go on through the kernels that walk whole frontiers
                hipLaunchKernelGGL(route_materialise_kernel, dim3(nq), dim3(256), seg_stride(pl.probes) * 4, s, v, pl.seg[cur], pl.probes,
                                   pl.rflag, pl.small_flag, pl.frontier[cur], pl.fstride, pl.fcount[cur]);
        }
        if (use_gemm) {
            const bool half = half_on(l);
            used_half |= half;
            const uint32_t n = g.n, kc = std::min(n, half ? pl.kc_half : pl.kc_split);
            launch_level_gemm(ix, pl, g, dq, nq, s, half);
            if (n <= 512 * 32)
                launch_prefilter(pl, nq, n, kc, nullptr, nullptr, nullptr, s);
            else
                hipLaunchKernelGGL(route_prefilter_kernel, dim3(nq), dim3(256), 0, s, pl.fdist, pl.fstride, n, kc, pl.rc_pos,
                                   pl.kc_cap, pl.rc_max);
            // (candidates by eighths of the level, an eighth per XCD: see the kernel; worth it once the level's rows are
            // more than one L2 holds anyway)
            static const int bucket_env = getenv("CUDDLE_REFINE_BUCKETS") ? atoi(getenv("CUDDLE_REFINE_BUCKETS")) : -1; // A/B aid: 0 off, Z
            const uint64_t table_bytes = (uint64_t)n * v.dim * 4;
            const bool buckets = bucket_env != 0 && n >= 512 && (bucket_env > 0 || (v.dim >= 512 && table_bytes > (16ull << 20)));
            const uint32_t bz = bucket_env > 0 ? (uint32_t)bucket_env : 1; // (parts of an eighth, one after the other: 2 and 4 measured slower)
            hipLaunchKernelGGL(route_refine_kernel, dim3(buckets ? 8 : (kc + 63) / 64, nq, buckets ? bz : 1), dim3(256),
                               (((size_t)v.dim + 3) & ~(size_t)3) * sizeof(float) + (buckets ? (size_t)kc * 4 : 0), s, v, dq,
                               g.nodes, pl.rc_pos, pl.kc_cap, kc, pl.rc_dist, pl.qnorm, 0u, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, zero_cnt, (ix->level_width[l] << po.shift) + 1,
                               buckets ? (n + 8 * bz - 1) / (8 * bz) : 0u);
            hipLaunchKernelGGL(route_finish_kernel, dim3(nq), dim3(256), route_finish_lds_bytes(host_next_pow2(kc), pl.probes), s,
                               v, g.nodes, n, pl.rc_pos, pl.rc_dist, pl.kc_cap, kc, pl.rc_max, pl.qsq, g.max_sqn, pl.probes,
                               pl.frontier[cur ^ 1], pl.fstride, pl.fcount[cur ^ 1], pl.leaves, pl.lstride, pl.lcount,
                               pl.rflag, &ix->d_stats->fallbacks, po, nullptr, nullptr, nullptr, nullptr, half ? 1u : 0u,
                               half ? ix->d_half_fb + l : nullptr);
            only = pl.rflag; // queries whose selection could not be proven exact fall through to the exact kernels
        }
        // blocks per query of the exact distance kernel (they stride over the frontier): after a filter path only the
        // (rare) flagged queries do work there; after the exact grouped path only the small frontiers; a per-query
        // level that was not worth grouping has, typically, frontiers of fewer than 256 nodes whatever the bound says
        // (and blocks that find nothing to do still cost their dispatch); a shared level's width is known
        // (a per-query level that was not worth grouping: about 8192 blocks in all - 8 per query for a batch of 1000,
        // 2 for one of 4096, where dispatching 32 768 mostly idle blocks cost more than the few wide frontiers gain)
        const uint32_t small_gx = std::max<uint32_t>(2, std::min<uint32_t>(8, 8192 / std::max<uint32_t>(nq, 1)));
        const uint32_t dist_gx = (use_gemm || grouped_filter) ? 1 : grouped ? 4
                                 : std::min<uint32_t>((bound + 63) / 64, (shared || (flags & CUDDLE_SEARCH_ROUTE_EXACT_ONLY)) ? 64 : small_gx);
        // after a filter path the exact kernels serve only the queries the filter left (normally none): one block per
        // query would compute their distances, so the selection kernel does it itself and the launch is saved
        const bool fuse_dist = !keep_all && (use_gemm || grouped_filter) && fuse_ok;
        FusedDist fd_args;
        if (fuse_dist) {
            fd_args.queries = dq;
            fd_args.qnorm = pl.qnorm;
            fd_args.fdist = pl.fdist;
            if (materialise_pending) { // ... and writes out the frontier of a query that only has its segment table
                fd_args.seg = pl.seg[cur];
                fd_args.small = pl.small_flag;
                fd_args.frontier = pl.frontier[cur];
                fd_args.pstride = pl.probes;
            }
        }
        // fewer probes than roots: every query measures the same few centroids, which then come from LDS
        const bool roots_path = !keep_all && !fuse_dist && sfr && !stats && !only && sn >= 4 && sn <= 100 &&
                                route_dist_roots_lds_bytes(v.dim, sn) <= 80 * 1024;
        if (roots_path)
            hipLaunchKernelGGL(route_dist_roots_kernel, dim3(std::min<uint32_t>((nq + kRootsWaves - 1) / kRootsWaves, 512)),
                               dim3(64 * kRootsWaves), route_dist_roots_lds_bytes(v.dim, sn), s, v, dq, nq, sfr, sn,
                               route_dist_roots_stride(v.dim), pl.fdist, pl.fstride);
        else if (!keep_all && !fuse_dist)
            hipLaunchKernelGGL(route_dist_kernel, dim3(dist_gx, nq), dim3(256), v.dim * sizeof(float), s, v, dq,
                               pl.frontier[cur], pl.fstride, pl.fcount[cur], sfr, sn, pl.fdist, pl.probes, stats, only, pl.qnorm);
        const uint32_t *sel_only = (grouped && !grouped_filter) ? nullptr : only; // after the exact grouped path every query still selects here
#define CUDDLE_SELECT_REG(T, K)                                                                                              \
    hipLaunchKernelGGL((route_select_reg_kernel<T, K>), dim3(nq), dim3(T), route_select_reg_lds_bytes(scap, fuse_dist ? v.dim : 0), s, v, \
                       pl.frontier[cur], pl.fstride, pl.fcount[cur], sfr, sn, pl.fdist, pl.probes, stats, scap,                  \
                       next_shared ? nullptr : pl.frontier[cur ^ 1], pl.fcount[cur ^ 1], pl.leaves, pl.lstride, pl.lcount,       \
                       sel_only, po, fd_args)
        // small blocks, as few keys per thread as the level's bound allows: a block's time is mostly
        // latency (loads, barriers), so what counts is how many of them are in flight
        if (!keep_all && bound <= 256 * 8)
            CUDDLE_SELECT_REG(256, 8);
        else if (!keep_all && bound <= 256 * 16)
            CUDDLE_SELECT_REG(256, 16);
        else if (!keep_all && bound <= 256 * 24)
            CUDDLE_SELECT_REG(256, 24);
        else if (!keep_all && bound <= 256 * 32)
            CUDDLE_SELECT_REG(256, 32);
        else if (!keep_all && bound <= 512 * 20)
            CUDDLE_SELECT_REG(512, 20);
        else if (!keep_all && bound <= 512 * 24)
            CUDDLE_SELECT_REG(512, 24);
        else if (!keep_all && bound <= 512 * 32)
            CUDDLE_SELECT_REG(512, 32);
        else
            hipLaunchKernelGGL(route_select_kernel, dim3(nq), dim3(256), route_select_lds_bytes(scap), s, v, pl.frontier[cur],
                               pl.fstride, pl.fcount[cur], sfr, sn, pl.fdist, pl.probes, stats, scap,
                               next_shared ? nullptr : pl.frontier[cur ^ 1], pl.fcount[cur ^ 1], pl.leaves, pl.lstride,
                               pl.lcount, sel_only, po);
        cur ^= 1;
        shared = next_shared;
        grouped = group_next;
        have_seg = seg_next;
        fb = fb_next;
    }
    if (stats) // what is left below the last level: empty, the frontier on which the reference's loop ends
        HIP_TRY(hipMemcpyAsync(pl.level_fc + (size_t)ix->n_levels * pl.qc, pl.fcount[cur], (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    if (used_half && feedback) half_feedback_post(ix, nq, s);
    return CUDDLE_OK;
}

#ifdef CUDDLE_STAMP_KERNEL
extern "C" int cuddle_debug_stamps(uint64_t *out, uint32_t n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_stamps), (size_t)std::min(n_blocks, kStampBlocks) * 64);
}
#endif

// Scan + phase-1 selection for the leaf lists in pl.leaves / pl.lcount.
// ghist != null: histogram mode (distances to the spill buffer, local histograms out, no selection).
static cuddle_status launch_scan_select(cuddle_index *ix, const Plan &pl, uint32_t nq, hipStream_t s,
                                        uint32_t *ghist = nullptr) {
    const IndexDev &v = ix->dev;
    // candidate counts, then every query's leaf tables written in block order, longest queries first
    // (nq <= kMaxChunk = 4096)
    mark(ix, s, CUDDLE_STAGE_SCAN_PREP);
    const uint32_t tstride = pl.lstride + 4;
    const bool by_leaf = pl.leaf_grouped && v.code_kind == CUDDLE_CODE_BINARY;
    // heaviest items first when leaves are reached often enough for items of very different sizes (where a leaf sees
    // two or three queries the items are all small, and leaf order keeps neighbouring leaves' codes together in time)
    const bool order_items = by_leaf && (uint64_t)nq * pl.lstride >= 8ull * std::max<uint32_t>(ix->n_leaves, 1);
    uint32_t *work_bins = pl.rq_cnt + v.n_nodes + 1;
    // The per-query scan's blocks: longest first (the tail of the launch is then made of short ones), or, where the code
    // array is too large for the L2s to hold it whatever the order, in the order of the queries' first leaves, XCD by
    // XCD, so that the queries that share leaves share an L2 (12.5M x 128-d: scan 0.113 -> 0.100 ms; 10M x 960-d query
    // by query 0.367 -> 0.317; RaBitQ 10M x 768-d 1.52 -> 1.46; 1M x 128-d, 16 MB of codes: 0.276 -> 0.291, hence the size)
    const uint64_t code_bytes = v.code_kind == CUDDLE_CODE_RABITQ ? (uint64_t)v.n_vectors * v.words32 * 4 : (uint64_t)v.n_vectors * v.words * 8;
    bool by_key = !by_leaf && code_bytes > (64ull << 20);
    if (const char *e = getenv("CUDDLE_SCAN_ORDER")) by_key = !by_leaf && atoi(e) == 1; // A/B aid: 0 = by work, 1 = by key
    hipLaunchKernelGGL(cand_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, v, pl.leaves, pl.lstride, pl.lcount, nq,
                       pl.cand_count, by_leaf ? pl.rq_cnt : (uint32_t *)nullptr, v.n_nodes + 1 + 2 * kWorkBins, pl.loc_key);
    hipLaunchKernelGGL(scan_tables_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, v, pl.leaves, pl.lstride, pl.lcount,
                       pl.cand_count, nq, pl.leaf_tables, tstride, by_leaf ? pl.rq_cnt : (uint32_t *)nullptr,
                       reinterpret_cast<uint2 *>(pl.rq_pinfo), pl.rq_pnode, pl.rq_pslot, pl.loc_key, pl.rr_order, by_key ? 1u : 0u);
    if (by_leaf) { // the pairs in leaf order, cut into work items of at most 64 queries
        const uint32_t n = v.n_nodes, nb = (n + kPairScanChunk - 1) / kPairScanChunk;
        const bool self_carry = nb <= 16;
        if (!self_carry) {
            hipLaunchKernelGGL(pair_block_sums_kernel, dim3(nb), dim3(256), 0, s, pl.rq_cnt, n, pl.rq_bsum, nb);
            hipLaunchKernelGGL(pair_block_scan_kernel, dim3(1), dim3(1024), 0, s, pl.rq_bsum, nb, n, pl.rq_poff, pl.rq_ioff);
        }
        hipLaunchKernelGGL(pair_scan_kernel, dim3(nb), dim3(256), 0, s, pl.rq_cnt, n, self_carry ? nullptr : pl.rq_bsum, nb, pl.rq_poff,
                           pl.rq_ioff, reinterpret_cast<uint4 *>(pl.rq_items), (const uint2 *)nullptr,
                           order_items ? work_bins : (uint32_t *)nullptr, v.vec_off);
        hipLaunchKernelGGL(leaf_pair_fill_kernel, dim3((pl.lstride + 255) / 256, nq), dim3(256), 0, s,
                           reinterpret_cast<const uint2 *>(pl.rq_pinfo), pl.rq_pnode, pl.rq_pslot, pl.lcount, pl.lstride, nq, pl.rq_poff,
                           reinterpret_cast<uint2 *>(pl.rq_psorted));
    }
    mark(ix, s, CUDDLE_STAGE_SCAN);
    if (v.code_kind == CUDDLE_CODE_RABITQ) {
        static const bool rabitq_exact_all = getenv("CUDDLE_RABITQ_EXACT_ALL") != nullptr; // A/B aid: no bounds, every estimate exact
        const float *rq = reinterpret_cast<const float *>(pl.qcodes), *qn = rq + (size_t)pl.qc * v.dim;
        // bound tiles cut every 64 scan positions across leaf ends (lanes address the code array by 32-bit word offsets)
        const char *pe = getenv("CUDDLE_RQ_PACK"); // A/B and test aid, read at every call: 0 = tiles cut at leaf ends
        const bool pack_env = !(pe && atoi(pe) == 0);
        const bool pack_ok = pack_env && v.n_vectors * (uint64_t)v.words32 < (1ull << 32);
        uint32_t pre_bounds = 0;
        if (pl.rq_grouped && !rabitq_exact_all) {
            // the (query, leaf) pairs of the leaf lists in leaf order, cut into work items; then the bounds of every
            // pair's candidates on the matrix cores, written to the queries' spill rows
            const uint32_t n = v.n_nodes, total = nq * pl.lstride, nb = (n + kPairScanChunk - 1) / kPairScanChunk;
            HIP_TRY(hipMemsetAsync(pl.rq_cnt, 0, ((size_t)n + 1) * 4, s));
            hipLaunchKernelGGL(rq_pairs_kernel, dim3((pl.lstride + 255) / 256, nq), dim3(256), 0, s, pl.leaf_tables, tstride, pl.leaves,
                               pl.lstride, pl.rq_cnt, reinterpret_cast<uint2 *>(pl.rq_pinfo), pl.rq_pnode, pl.rq_pslot);
            const bool self_carry = nb <= 16;
            if (!self_carry) {
                hipLaunchKernelGGL(pair_block_sums_kernel, dim3(nb), dim3(256), 0, s, pl.rq_cnt, n, pl.rq_bsum, nb);
                hipLaunchKernelGGL(pair_block_scan_kernel, dim3(1), dim3(1024), 0, s, pl.rq_bsum, nb, n, pl.rq_poff, pl.rq_ioff);
            }
            hipLaunchKernelGGL(pair_scan_kernel, dim3(nb), dim3(256), 0, s, pl.rq_cnt, n, self_carry ? nullptr : pl.rq_bsum, nb, pl.rq_poff,
                               pl.rq_ioff, reinterpret_cast<uint4 *>(pl.rq_items), (const uint2 *)nullptr);
            hipLaunchKernelGGL(rq_pair_fill_kernel, dim3((total + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint2 *>(pl.rq_pinfo),
                               pl.rq_pnode, pl.rq_pslot, total, pl.rq_poff, reinterpret_cast<uint2 *>(pl.rq_psorted));
            const uint32_t max_items = total / kGroupPairs + std::min(n, total) + 1;
            static const int mt = getenv("CUDDLE_RQ_MT") ? atoi(getenv("CUDDLE_RQ_MT")) : 8; // tuning aid: 16-vector tiles per pass
#define CUDDLE_RQ_GROUPED(MT)                                                                                                 \
    hipLaunchKernelGGL(rq_bounds_grouped_kernel<MT>, dim3((max_items + 3) / 4), dim3(256), 0, s, v, pl.rq_ioff + n,              \
                       reinterpret_cast<const uint4 *>(pl.rq_items), reinterpret_cast<const uint2 *>(pl.rq_psorted), pl.rq_i8,  \
                       pl.rq_rows, qn, reinterpret_cast<float *>(pl.gdist), pl.cstride)
            if (mt == 4) CUDDLE_RQ_GROUPED(4); else if (mt == 2) CUDDLE_RQ_GROUPED(2); else CUDDLE_RQ_GROUPED(8);
#undef CUDDLE_RQ_GROUPED
            pre_bounds = rq_grouped_env() == 2 ? 2u : 1u;
        }
        hipLaunchKernelGGL(scan_select_rabitq_kernel, dim3(nq), dim3(pl.scan_threads), pl.scan_lds, s, v, rq, qn, pl.leaf_tables,
                           tstride, pl.lstride, pl.tcap, pl.rerank_k, pl.dist_cap, reinterpret_cast<float *>(pl.gdist),
                           pl.cstride, pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count, pl.cand_count, rabitq_exact_all ? 1u : 0u,
                           &ix->d_stats->rq_exact, pl.rq_rows, pre_bounds, pack_ok ? 1u : 0u);
        if (pre_bounds == 2) { // checking mode: any bound that differs from the per-query kernel's fails the search
            DevStats h;
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(&h, ix->d_stats, sizeof(h), hipMemcpyDeviceToHost));
            if (h.rq_mismatch) return fail(CUDDLE_ERR_HIP, "%llu RaBitQ bounds of the grouped kernel differ from the per-query kernel's", h.rq_mismatch);
        }
        hipLaunchKernelGGL(selection_sort_f32_kernel, dim3(nq), dim3(256), (size_t)host_next_pow2(pl.sstride) * 8, s,
                           pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count, pl.sel_ord);
        HIP_TRY(hipGetLastError());
        return CUDDLE_OK;
    }
    if (by_leaf) {
        const uint32_t total = nq * pl.lstride, max_items = total / kGroupPairs + std::min(v.n_nodes, total) + 1;
        const bool order_items = (uint64_t)nq * pl.lstride >= 8ull * std::max<uint32_t>(ix->n_leaves, 1); // (as above)
        const uint32_t *order = order_items ? pl.rq_order : nullptr;
        if (order_items)
            hipLaunchKernelGGL(item_scatter_kernel, dim3((max_items + 255) / 256), dim3(256), 0, s, v, pl.rq_ioff + v.n_nodes,
                               reinterpret_cast<const uint4 *>(pl.rq_items), pl.rq_cnt + v.n_nodes + 1, pl.rq_order);
        if (v.dim < 255)
            hipLaunchKernelGGL(leaf_kernel8_for(v.words), dim3(max_items), dim3(128), 0, s, v, pl.rq_ioff + v.n_nodes,
                               reinterpret_cast<const uint4 *>(pl.rq_items), order, reinterpret_cast<const uint2 *>(pl.rq_psorted),
                               pl.qcodes, reinterpret_cast<uint8_t *>(pl.gdist), pl.cstride);
        else
            hipLaunchKernelGGL(leaf_kernel16_for(v.words), dim3(max_items), dim3(128), 0, s, v, pl.rq_ioff + v.n_nodes,
                               reinterpret_cast<const uint4 *>(pl.rq_items), order, reinterpret_cast<const uint2 *>(pl.rq_psorted),
                               pl.qcodes, reinterpret_cast<uint16_t *>(pl.gdist), pl.cstride);
    }
    if (by_leaf) {
        // the distances are in the spill rows: histogram, threshold and selection from there, every query of the chunk resident
        const size_t lds = select_rows_lds_bytes(v.dim, pl.lstride);
        if (v.dim < 255)
            hipLaunchKernelGGL(select_rows_kernel<uint8_t>, dim3(nq), dim3(512), lds, s, v, pl.leaf_tables, tstride, pl.lstride, pl.rerank_k,
                               reinterpret_cast<const uint8_t *>(pl.gdist), pl.cstride, pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count,
                               pl.cand_count, ghist);
        else
            hipLaunchKernelGGL(select_rows_kernel<uint16_t>, dim3(nq), dim3(512), lds, s, v, pl.leaf_tables, tstride, pl.lstride, pl.rerank_k,
                               reinterpret_cast<const uint16_t *>(pl.gdist), pl.cstride, pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count,
                               pl.cand_count, ghist);
    } else {
        hipLaunchKernelGGL(scan_kernel_for(v.dim, v.words), dim3(nq), dim3(pl.scan_threads), pl.scan_lds, s, v, pl.qcodes,
                           pl.leaf_tables, tstride, pl.lstride, pl.tcap, pl.rerank_k, pl.dist_cap, (void *)pl.gdist, pl.cstride,
                           pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count, pl.cand_count, ghist);
    }
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

// in_key_order: the queries are taken in the order launch_scan_select left in pl.rr_order for these nq queries.
static cuddle_status launch_rerank(cuddle_index *ix, const Plan &pl, const float *dq, uint32_t nq, hipStream_t s,
                                   bool in_key_order = true) {
    const IndexDev &v = ix->dev;
    launch_query_norms(ix, pl, dq, nq, s);
    static const bool no_order = getenv("CUDDLE_RERANK_ORDER") && atoi(getenv("CUDDLE_RERANK_ORDER")) == 0; // A/B aid
    const uint32_t *order = in_key_order && !no_order ? pl.rr_order : nullptr;
    hipLaunchKernelGGL(rerank_kernel, dim3((pl.sstride + 63) / 64, order ? (nq + 7) & ~7u : nq), dim3(256), v.dim * sizeof(float), s, v, dq,
                       pl.sel_pos, pl.sstride, pl.sel_count, pl.sel_dist, pl.qnorm, order, nq);
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

// lazy_stats: the per-query counts are summed when the statistics are read (single-chunk searches on the handle's own
// work space) instead of by a launch here.
static cuddle_status launch_topk(cuddle_index *ix, const Plan &pl, uint32_t nq, uint64_t *d_ids, float *d_dist,
                                 uint32_t *d_count, hipStream_t s, uint64_t *d_order = nullptr, bool lazy_stats = false) {
    const IndexDev &v = ix->dev;
    const uint32_t rk = v.code_kind == CUDDLE_CODE_RABITQ ? 1u : 0u;
#define CUDDLE_TOPK_WAVE(KPL)                                                                                                 \
    hipLaunchKernelGGL(topk_wave_kernel<KPL>, dim3(nq), dim3(64), 0, s, v, pl.sel_pos, pl.sel_key, pl.sel_dist,                \
                       pl.sstride, pl.sel_count, pl.k, rk, d_ids, d_dist, d_count, d_order ? pl.sel_ord : nullptr, d_order, nq)
    if (pl.k <= 16 && pl.sstride <= 256) {
        CUDDLE_TOPK_WAVE(4);
    } else if (pl.k <= 16 && pl.sstride <= 512) {
        CUDDLE_TOPK_WAVE(8);
    } else if (pl.k <= 16 && pl.sstride <= 1024) {
        CUDDLE_TOPK_WAVE(16);
    } else if (pl.k <= 16 && pl.sstride <= 2048) { // (10 of 1500: 0.45 ms as a block sort of 2048 keys per query, 10 000 queries)
        CUDDLE_TOPK_WAVE(32);
    } else {
        size_t lds = (size_t)host_next_pow2(pl.sstride) * 8;
        hipLaunchKernelGGL(topk_kernel, dim3(nq), dim3(256), lds, s, v, pl.sel_pos, pl.sel_key, pl.sel_dist, pl.sstride,
                           pl.sel_count, pl.k, rk, d_ids, d_dist, d_count, d_order ? pl.sel_ord : nullptr, d_order);
    }
    if (lazy_stats) {
        ix->lazy_lcount = pl.lcount; ix->lazy_cand = pl.cand_count; ix->lazy_sel = pl.sel_count;
        ix->lazy_nq = nq;
        ix->lazy_pending = true;
    } else {
        hipLaunchKernelGGL(stats_accum_kernel, dim3(std::min<uint32_t>((nq + 255) / 256, 8)), dim3(256), 0, s, pl.lcount, pl.cand_count,
                           pl.sel_count, nq, ix->d_stats);
    }
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

// The whole path on device pointers; chunks the batch so the work space stays bounded.
// fixed_ws != null: the work space is that buffer of fixed_cap bytes and nothing is allocated (no stage timing either:
// its events are created on demand).
static cuddle_status search_device_locked(cuddle_index *ix, const float *dq, uint32_t nq, uint32_t dim, uint32_t k,
                                          uint32_t probes, uint32_t rf, uint32_t flags, uint64_t *d_ids, float *d_dist,
                                          uint32_t *d_count, hipStream_t s, void *fixed_ws = nullptr, size_t fixed_cap = 0) {
    if (!dq || !d_ids || !d_dist) return fail(CUDDLE_ERR_BAD_ARG, "null query or output pointer");
    if (dim != ix->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, ix->dev.dim);
    if (probes == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    uint64_t rerank_k;
    ST_TRY(rerank_k_of(k, rf, flags, rerank_k));
    ix->timed = !fixed_ws && (flags & CUDDLE_SEARCH_TIME_STAGES) != 0 && ix->events_used + 4096 < kMaxEvents;
    if (!fixed_ws) ix->shard_state = 0; // the work space is about to be re-planned: an unfinished sharded batch is void
    ST_TRY(enter_stream(ix, s));
    ix->last_levels = ix->n_levels;
    if (nq == 0) {
        HIP_TRY(hipMemsetAsync(ix->d_stats, 0, sizeof(DevStats), s));
        return CUDDLE_OK;
    }
    // Two lanes. A batch larger than one chunk used to run its chunks one after the other through the one work space:
    // while a chunk selects, sorts or waits for the next of its thirty dependent launches, the memory system idles.
    // Such a batch is now cut into an even number of equal pieces that alternate between two streams of the handle's
    // own, each with its own half of the work space; the caller's stream forks into them and joins them again (10 000
    // 128-d queries: 1.51 -> 1.39 ms). Every query's result is what the single lane gives (pieces are independent
    // searches). A batch that fits one chunk stays whole: cut in two it runs slower (1000 960-d queries 0.68 -> 0.72 ms,
    // 4096 128-d queries 0.85 -> 0.90 ms), the kernels of a half batch fill the chip worse than overlap wins back.
    // Not when stages are timed (events on one stream), under graph capture, or on the fixed work space.
    static const int lanes_env = getenv("CUDDLE_LANES") ? atoi(getenv("CUDDLE_LANES")) : -1; // tuning aid: 1 = one lane always
    if (!fixed_ws && !ix->timed && lanes_env != 1 && (nq > kMaxChunk || lanes_env == 2) && !stream_capturing(s)) {
        const uint32_t n_pieces = 2 * ((nq + 2 * kMaxChunk - 1) / (2 * kMaxChunk)), piece = (nq + n_pieces - 1) / n_pieces;
        Plan lane[2];
        ST_TRY(make_plan(ix, piece, k, probes, rerank_k, 0, lane[0]));
        if (lane[0].qc >= piece) { // (a piece that must itself be cut to fit the budget: the plain loop below)
            const size_t half = align_up(lane[0].bytes, 256);
            ST_TRY(ix->ws.ensure(2 * half));
            lane[1] = lane[0];
            carve_plan(ix, lane[0], static_cast<char *>(ix->ws.p));
            carve_plan(ix, lane[1], static_cast<char *>(ix->ws.p) + half);
            HIP_TRY(hipMemsetAsync(ix->d_stats, 0, sizeof(DevStats), s));
            HIP_TRY(hipEventRecord(ix->fork_ev, s));
            for (int l = 0; l < 2; ++l) HIP_TRY(hipStreamWaitEvent(ix->lane_stream[l], ix->fork_ev, 0));
            uint32_t pi = 0;
            for (uint32_t q0 = 0; q0 < nq; q0 += piece, ++pi) {
                const uint32_t qc = std::min(piece, nq - q0);
                const float *q = dq + (uint64_t)q0 * dim;
                const Plan &pl = lane[pi & 1];
                hipStream_t ls = ix->lane_stream[pi & 1];
                bool init_done = false;
                ST_TRY(launch_quantize(ix, pl, q, qc, ls, &init_done, flags, false));
                ST_TRY(launch_route(ix, pl, q, qc, flags, ls, pi == 0, false, init_done));
                ST_TRY(launch_scan_select(ix, pl, qc, ls));
                ST_TRY(launch_rerank(ix, pl, q, qc, ls));
                ST_TRY(launch_topk(ix, pl, qc, d_ids + (uint64_t)q0 * k, d_dist + (uint64_t)q0 * k, d_count ? d_count + q0 : nullptr, ls));
            }
            for (int l = 0; l < 2; ++l) {
                HIP_TRY(hipEventRecord(ix->lane_ev[l], ix->lane_stream[l]));
                HIP_TRY(hipStreamWaitEvent(s, ix->lane_ev[l], 0));
            }
            return CUDDLE_OK;
        }
    }
    Plan pl;
    ST_TRY(make_plan(ix, nq, k, probes, rerank_k, 0, pl, fixed_ws, fixed_cap));
    if (ix->timed) ix->timed_searches++;
    for (uint32_t q0 = 0; q0 < nq; q0 += pl.qc) {
        const uint32_t qc = std::min(pl.qc, nq - q0);
        const float *q = dq + (uint64_t)q0 * dim;
        mark(ix, s, CUDDLE_STAGE_QUANTIZE);
        bool init_done = false;
        ST_TRY(launch_quantize(ix, pl, q, qc, s, &init_done, flags, q0 == 0)); // (the first chunk's first launch clears the counters)
        mark(ix, s, CUDDLE_STAGE_ROUTE);
        ST_TRY(launch_route(ix, pl, q, qc, flags, s, fixed_ws == nullptr, q0 == 0 && !init_done, init_done));
        mark(ix, s, CUDDLE_STAGE_SCAN);
        ST_TRY(launch_scan_select(ix, pl, qc, s));
        mark(ix, s, CUDDLE_STAGE_RERANK);
        ST_TRY(launch_rerank(ix, pl, q, qc, s));
        mark(ix, s, CUDDLE_STAGE_TOPK);
        ST_TRY(launch_topk(ix, pl, qc, d_ids + (uint64_t)q0 * k, d_dist + (uint64_t)q0 * k, d_count ? d_count + q0 : nullptr, s,
                           nullptr, !fixed_ws && nq <= pl.qc));
        mark(ix, s, -1);
    }
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_search_batch_device(cuddle_index *idx, const float *d_queries, uint32_t n_queries,
                                                    uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                                    uint32_t rerank_factor, uint32_t flags, uint64_t *d_out_ids,
                                                    float *d_out_dist, uint32_t *d_out_count, void *stream) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    return search_device_locked(idx, d_queries, n_queries, dim, k, probes_per_level, rerank_factor, flags, d_out_ids,
                                d_out_dist, d_out_count, static_cast<hipStream_t>(stream));
}

extern "C" cuddle_status cuddle_search_batch_device_noalloc(cuddle_index *idx, const float *d_queries, uint32_t n_queries,
                                                            uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                                            uint32_t rerank_factor, uint32_t flags, uint64_t *d_out_ids,
                                                            float *d_out_dist, uint32_t *d_out_count, void *stream) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (!idx->scratch_ptr) return fail(CUDDLE_ERR_HIP, "the fixed work space was not resolved on device %d", idx->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = idx->device;
    // The array is shared by every handle on this GPU: one enqueue at a time, and a call on another stream than the
    // last one is ordered behind that stream on the device. (Under graph capture nothing is awaited or recorded.)
    std::lock_guard<std::mutex> lk2(g_scratch_call_mu[d]);
    const bool capturing = stream_capturing(s);
    if (!capturing) {
        if (g_scratch_used[d] && g_scratch_stream[d] != s) order_after(g_scratch_stream[d], s, g_scratch_ev[d]);
        g_scratch_stream[d] = s;
        g_scratch_used[d] = true;
    }
    return search_device_locked(idx, d_queries, n_queries, dim, k, probes_per_level, rerank_factor, flags, d_out_ids, d_out_dist,
                                d_out_count, s, idx->scratch_ptr, kScratchBytes);
}

static cuddle_status read_stats(cuddle_index *ix, cuddle_stats *st) {
    DevStats h;
    flush_lazy_stats(ix); // the last search left its per-query counts in the work space: sum them now
    if (hipStreamSynchronize(ix->last_stream) != hipSuccess) { // the caller destroyed that stream: wait for the device
        (void)hipGetLastError();
        HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipMemcpy(&h, ix->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    st->leaves_searched = h.leaves;
    st->vectors_scanned = h.scanned;
    st->vectors_reranked = h.reranked;
    st->max_candidates = h.max_cand;
    st->max_leaves = h.max_leaves;
    st->tree_levels = ix->last_levels;
    st->route_fallbacks = h.fallbacks;
    st->rabitq_exact_estimates = h.rq_exact;
    st->rabitq_exact_all_queries = h.rq_exact_all;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_last_stats(cuddle_index *idx, cuddle_stats *stats) {
    if (!idx || !stats) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    return read_stats(idx, stats);
}

extern "C" cuddle_status cuddle_stage_times(cuddle_index *idx, float *ms, uint32_t *n_searches) {
    if (!idx || !ms) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    for (int i = 0; i < CUDDLE_STAGE_COUNT; ++i) ms[i] = 0.f;
    if (n_searches) *n_searches = idx->timed_searches;
    idx->timed_searches = 0;
    if (idx->events_used == 0) return CUDDLE_OK;
    const size_t used = idx->events_used;
    idx->events_used = 0;
    HIP_TRY(hipEventSynchronize(idx->events[used - 1]));
    for (size_t c = 0; c + 1 < used; ++c) {
        const int st = idx->event_stage[c];
        if (st < 0 || st >= CUDDLE_STAGE_COUNT) continue;
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, idx->events[c], idx->events[c + 1]));
        ms[st] += t;
    }
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_index_reserve(cuddle_index *idx, uint32_t max_queries, uint32_t max_k,
                                              uint32_t max_probes, uint32_t max_rerank_k) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    Plan pl;
    return make_plan(idx, std::max(max_queries, 1u), std::max(max_k, 1u), std::max(max_probes, 1u),
                     std::max<uint64_t>(max_rerank_k, 1), 0, pl);
}

// Staging buffer of the host entry points: [queries][ids][dist][count] for one call.
struct HostIO {
    float *q = nullptr;
    uint64_t *ids = nullptr;
    float *dist = nullptr;
    uint32_t *count = nullptr;
};

static cuddle_status carve_io(cuddle_index *ix, uint32_t nq, uint32_t dim, uint32_t k, size_t extra, HostIO &io, char **extra_p) {
    size_t o_q = 0, o_ids = align_up(o_q + (size_t)nq * dim * 4, 256), o_dist = align_up(o_ids + (size_t)nq * k * 8, 256);
    size_t o_cnt = align_up(o_dist + (size_t)nq * k * 4, 256), o_extra = align_up(o_cnt + (size_t)nq * 4, 256);
    ST_TRY(ix->io.ensure(o_extra + extra + 256));
    char *b = static_cast<char *>(ix->io.p);
    io.q = (float *)(b + o_q);
    io.ids = (uint64_t *)(b + o_ids);
    io.dist = (float *)(b + o_dist);
    io.count = (uint32_t *)(b + o_cnt);
    if (extra_p) *extra_p = b + o_extra;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_search_batch(cuddle_index *idx, const float *queries, uint32_t n_queries, uint32_t dim,
                                             uint32_t k, uint32_t probes_per_level, uint32_t rerank_factor,
                                             uint32_t flags, uint64_t *out_ids, float *out_dist, uint32_t *out_count,
                                             cuddle_stats *stats) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !out_ids || !out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null query or output pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (k == 0) return fail(CUDDLE_ERR_BAD_ARG, "k must be >= 1 (the reference underflows k-1, hierarchical.rs:651)");
    HostIO io;
    ST_TRY(carve_io(idx, std::max(n_queries, 1u), dim, k, 0, io, nullptr));
    if (n_queries) HIP_TRY(hipMemcpy(io.q, queries, (size_t)n_queries * dim * 4, hipMemcpyHostToDevice));
    ST_TRY(search_device_locked(idx, io.q, n_queries, dim, k, probes_per_level, rerank_factor, flags, io.ids, io.dist,
                                io.count, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (n_queries) {
        HIP_TRY(hipMemcpy(out_ids, io.ids, (size_t)n_queries * k * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_dist, io.dist, (size_t)n_queries * k * 4, hipMemcpyDeviceToHost));
        if (out_count) HIP_TRY(hipMemcpy(out_count, io.count, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
    }
    if (stats) ST_TRY(read_stats(idx, stats));
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_search_batch_with_stats(cuddle_index *idx, const float *queries, uint32_t n_queries,
                                                        uint32_t dim, uint32_t k, uint32_t probes, uint32_t rerank_factor,
                                                        uint64_t *out_ids, float *out_dist, uint32_t *out_count,
                                                        uint32_t *out_leaves, uint32_t leaf_stride, uint32_t *out_leaf_count,
                                                        uint32_t *out_level_nodes, uint32_t level_stride,
                                                        uint32_t *out_level_count, uint32_t *out_scanned,
                                                        uint32_t *out_reranked) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !out_ids || !out_dist || !out_count || !out_leaves || !out_leaf_count || !out_level_nodes ||
        !out_level_count || !out_scanned || !out_reranked)
        return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (probes == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes must be >= 1");
    const uint32_t flags = CUDDLE_SEARCH_STATS_VARIANT;
    uint64_t rerank_k;
    ST_TRY(rerank_k_of(k, rerank_factor, flags, rerank_k));
    if (level_stride < idx->n_levels + 1) return fail(CUDDLE_ERR_BAD_ARG, "level_stride %u < tree levels + 1 = %u", level_stride, idx->n_levels + 1);
    if (leaf_stride < plan_leaf_stride(idx, probes))
        return fail(CUDDLE_ERR_BAD_ARG, "leaf_stride %u < cuddle_index_leaf_stride() = %u", leaf_stride, plan_leaf_stride(idx, probes));
    ST_TRY(enter_stream(idx, nullptr));
    idx->timed = false;
    idx->shard_state = 0;
    idx->last_levels = idx->n_levels;
    HIP_TRY(hipMemsetAsync(idx->d_stats, 0, sizeof(DevStats), nullptr));
    if (n_queries == 0) return CUDDLE_OK;
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, k, probes, rerank_k, 0, pl));
    HostIO io;
    ST_TRY(carve_io(idx, pl.qc, dim, k, 0, io, nullptr));
    const uint32_t L = idx->n_levels;
    std::vector<uint32_t> lfc((size_t)pl.qc * (L + 1));
    for (uint32_t q0 = 0; q0 < n_queries; q0 += pl.qc) {
        const uint32_t qc = std::min(pl.qc, n_queries - q0);
        HIP_TRY(hipMemcpy(io.q, queries + (uint64_t)q0 * dim, (size_t)qc * dim * 4, hipMemcpyHostToDevice));
        ST_TRY(launch_quantize(idx, pl, io.q, qc, nullptr));
        ST_TRY(launch_route(idx, pl, io.q, qc, flags, nullptr));
        ST_TRY(launch_scan_select(idx, pl, qc, nullptr));
        ST_TRY(launch_rerank(idx, pl, io.q, qc, nullptr));
        ST_TRY(launch_topk(idx, pl, qc, io.ids, io.dist, io.count, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(out_ids + (uint64_t)q0 * k, io.ids, (size_t)qc * k * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_dist + (uint64_t)q0 * k, io.dist, (size_t)qc * k * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_count + q0, io.count, (size_t)qc * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_leaf_count + q0, pl.lcount, (size_t)qc * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy2D(out_leaves + (uint64_t)q0 * leaf_stride, (size_t)leaf_stride * 4, pl.leaves, (size_t)pl.lstride * 4,
                            (size_t)std::min(pl.lstride, leaf_stride) * 4, qc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_scanned + q0, pl.cand_count, (size_t)qc * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_reranked + q0, pl.sel_count, (size_t)qc * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy2D(lfc.data(), (size_t)qc * 4, pl.level_fc, (size_t)pl.qc * 4, (size_t)qc * 4, L + 1, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < qc; ++q) {
            // the reference pushes the frontier size at the top of every loop iteration and leaves the loop on
            // the first empty one (hierarchical.rs:860-865): the sizes up to and including the first zero
            uint32_t *row = out_level_nodes + (uint64_t)(q0 + q) * level_stride, n = 0;
            for (uint32_t l = 0; l <= L; ++l) {
                const uint32_t sz = lfc[(size_t)l * qc + q];
                row[n++] = sz;
                if (sz == 0) break;
            }
            out_level_count[q0 + q] = n;
        }
    }
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// stage entry points
// ---------------------------------------------------------------------------------------------
struct TmpDev { // scope-bound device allocation for the pair kernels
    void *p = nullptr;
    ~TmpDev() { if (p) (void)hipFree(p); }
    cuddle_status alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) { p = nullptr; return fail(CUDDLE_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        return CUDDLE_OK;
    }
};

extern "C" cuddle_status cuddle_distance_pairs(const float *a, const float *b, uint64_t n, uint32_t dim, uint32_t metric,
                                               float *out, int device_id) {
    if ((!a || !b || !out) && n) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (metric > CUDDLE_METRIC_DOT) return fail(CUDDLE_ERR_BAD_ARG, "unknown metric %u", metric);
    ST_TRY(use_device(device_id));
    if (n == 0) return CUDDLE_OK;
    TmpDev da, db, dout;
    size_t bytes = (size_t)n * dim * 4;
    ST_TRY(da.alloc(bytes)); ST_TRY(db.alloc(bytes)); ST_TRY(dout.alloc(n * 4));
    if (bytes) {
        HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    }
    uint64_t threads = n * 4;
    hipLaunchKernelGGL(distance_pairs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, (const float *)da.p,
                       (const float *)db.p, n, dim, metric, (float *)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_binary_quantize(const float *vectors, uint64_t n, uint32_t dim, float threshold,
                                                uint64_t *codes, int device_id) {
    if ((!vectors || !codes) && n) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim must be >= 1");
    ST_TRY(use_device(device_id));
    if (n == 0) return CUDDLE_OK;
    const uint32_t W = (dim + 63) / 64;
    TmpDev dv, dc;
    ST_TRY(dv.alloc((size_t)n * dim * 4)); ST_TRY(dc.alloc((size_t)n * W * 8));
    HIP_TRY(hipMemcpy(dv.p, vectors, (size_t)n * dim * 4, hipMemcpyHostToDevice));
    uint64_t total = n * W;
    hipLaunchKernelGGL(binary_quantize_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, 0, (const float *)dv.p, n,
                       dim, W, threshold, (uint64_t *)dc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(codes, dc.p, (size_t)n * W * 8, hipMemcpyDeviceToHost));
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_hamming_pairs(const uint64_t *a, const uint64_t *b, uint64_t n, uint32_t words,
                                              uint32_t *out, int device_id) {
    if ((!a || !b || !out) && n) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    ST_TRY(use_device(device_id));
    if (n == 0) return CUDDLE_OK;
    TmpDev da, db, dout;
    size_t bytes = (size_t)n * words * 8;
    ST_TRY(da.alloc(bytes)); ST_TRY(db.alloc(bytes)); ST_TRY(dout.alloc(n * 4));
    if (bytes) {
        HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(hamming_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint64_t *)da.p,
                       (const uint64_t *)db.p, n, words, (uint32_t *)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return CUDDLE_OK;
}

// BinaryQuantizer::from_vectors (binary.rs:122-141): sequential f32 fold, then / (count as f32).
extern "C" float cuddle_binary_threshold(const float *vectors, uint64_t n, uint32_t dim) {
    volatile float total = 0.0f; // volatile: keep the strictly sequential single-precision fold
    const uint64_t count = n * (uint64_t)dim;
    float t = 0.0f;
    for (uint64_t i = 0; i < count; ++i) t += vectors[i];
    total = t;
    return total / (float)count;
}

extern "C" cuddle_status cuddle_scan_mode(cuddle_index *idx, uint32_t n_queries, uint32_t probes_per_level, uint32_t *out_mode) {
    if (!idx || !out_mode) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    std::lock_guard<std::mutex> lk(idx->mu);
    const uint32_t lstride = plan_leaf_stride(idx, probes_per_level);
    const uint32_t piece = n_queries > kMaxChunk ? (n_queries + 2 * ((n_queries + 2 * kMaxChunk - 1) / (2 * kMaxChunk)) - 1) /
                                                       (2 * ((n_queries + 2 * kMaxChunk - 1) / (2 * kMaxChunk)))
                                                 : n_queries;
    *out_mode = idx->dev.code_kind == CUDDLE_CODE_BINARY && leaf_grouped_wanted(idx, std::max(piece, 1u), lstride) ? 1u : 0u;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_scan_read_ceiling(cuddle_index *idx, const float *d_queries, uint32_t n_batches,
                                                  uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                                  uint32_t rerank_factor, uint32_t reps, float *out_read_ms, float *out_scan_ms,
                                                  uint64_t *out_code_bytes) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!d_queries || !out_read_ms || !out_scan_ms || !out_code_bytes) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    const IndexDev &v = idx->dev;
    if (dim != v.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, v.dim);
    if (probes_per_level == 0 || reps == 0 || n_batches == 0 || n_batches > 64)
        return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level and reps must be >= 1, 1..64 batches");
    CeilingKernel ck = v.code_kind == CUDDLE_CODE_BINARY ? ceiling_kernel_for(v.words) : nullptr;
    if (!ck) return fail(CUDDLE_ERR_UNSUPPORTED, "no read-only twin for this code (binary codes of 2, 4, 8, 12, 15, 16 or 24 words)");
    uint64_t rerank_k;
    ST_TRY(rerank_k_of(k, rerank_factor, 0, rerank_k));
    ST_TRY(enter_stream(idx, nullptr));
    idx->timed = false;
    idx->shard_state = 0;
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, k, probes_per_level, rerank_k, 0, pl));
    if (n_queries == 0 || n_queries > pl.qc) return fail(CUDDLE_ERR_BAD_ARG, "1..%u queries per batch (one chunk)", pl.qc);
    (void)hipFuncSetAttribute((const void *)ck, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s = nullptr;
    // every batch is routed once; its blocks' tables and its query codes are kept aside, so that the timed launches can
    // go round the batches as a caller's successive searches would (a batch's leaves are cold again when its turn comes)
    const uint32_t tstride = pl.lstride + 4;
    const size_t tbytes = align_up((size_t)n_queries * 4 * tstride * 4, 256), qbytes = align_up((size_t)n_queries * v.words * 8, 256);
    TmpDev keep;
    ST_TRY(keep.alloc((tbytes + qbytes) * n_batches));
    char *kb = static_cast<char *>(keep.p);
    uint64_t total = 0;
    std::vector<uint32_t> cc(n_queries);
    for (uint32_t b = 0; b < n_batches; ++b) {
        const float *q = d_queries + (uint64_t)b * n_queries * dim;
        ST_TRY(launch_quantize(idx, pl, q, n_queries, s));
        ST_TRY(launch_route(idx, pl, q, n_queries, 0, s, false, true));
        ST_TRY(launch_scan_select(idx, pl, n_queries, s)); // also leaves the blocks' tables in pl.leaf_tables
        HIP_TRY(hipMemcpyAsync(kb + (tbytes + qbytes) * b, pl.leaf_tables, (size_t)n_queries * 4 * tstride * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(kb + (tbytes + qbytes) * b + tbytes, pl.qcodes, (size_t)n_queries * v.words * 8, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpy(cc.data(), pl.cand_count, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
        for (uint32_t c : cc) total += c;
    }
    hipEvent_t ev[3];
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    auto tables_of = [&](uint32_t b) { return reinterpret_cast<const uint32_t *>(kb + (tbytes + qbytes) * b); };
    auto codes_of = [&](uint32_t b) { return reinterpret_cast<const uint64_t *>(kb + (tbytes + qbytes) * b + tbytes); };
    auto read_once = [&](uint32_t b) {
        hipLaunchKernelGGL(ck, dim3(n_queries), dim3(pl.scan_threads), pl.scan_lds, s, v, tables_of(b), tstride, pl.lstride, pl.tcap, pl.sel_count);
    };
    auto scan_once = [&](uint32_t b) {
        hipLaunchKernelGGL(scan_kernel_for(v.dim, v.words), dim3(n_queries), dim3(pl.scan_threads), pl.scan_lds, s, v, codes_of(b),
                           tables_of(b), tstride, pl.lstride, pl.tcap, pl.rerank_k, pl.dist_cap, (void *)pl.gdist, pl.cstride,
                           pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count, pl.cand_count, (uint32_t *)nullptr);
    };
    auto body = [&]() -> cuddle_status {
        for (uint32_t b = 0; b < n_batches; ++b) read_once(b); // one untimed round
        HIP_TRY(hipEventRecord(ev[0], s));
        for (uint32_t r = 0; r < reps; ++r)
            for (uint32_t b = 0; b < n_batches; ++b) read_once(b);
        HIP_TRY(hipEventRecord(ev[1], s));
        for (uint32_t r = 0; r < reps; ++r)
            for (uint32_t b = 0; b < n_batches; ++b) scan_once(b);
        HIP_TRY(hipEventRecord(ev[2], s));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventSynchronize(ev[2]));
        float a = 0.f, bms = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&bms, ev[1], ev[2]));
        *out_read_ms = a / (float)(reps * n_batches);
        *out_scan_ms = bms / (float)(reps * n_batches);
        *out_code_bytes = total * 8ull * v.words / n_batches;
        return CUDDLE_OK;
    };
    const cuddle_status st = body();
    for (auto &e : ev) (void)hipEventDestroy(e);
    return st;
}

extern "C" cuddle_status cuddle_route_batch(cuddle_index *idx, const float *queries, uint32_t n_queries, uint32_t dim,
                                            uint32_t probes_per_level, uint32_t flags, uint32_t *out_leaves,
                                            uint32_t leaf_stride, uint32_t *out_leaf_count) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !out_leaves || !out_leaf_count) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    if (n_queries == 0) return CUDDLE_OK;
    ST_TRY(enter_stream(idx, nullptr));
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, 1, probes_per_level, 1, 0, pl));
    HostIO io;
    ST_TRY(carve_io(idx, pl.qc, dim, 1, 0, io, nullptr));
    idx->timed = false;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += pl.qc) {
        const uint32_t qc = std::min(pl.qc, n_queries - q0);
        HIP_TRY(hipMemcpy(io.q, queries + (uint64_t)q0 * dim, (size_t)qc * dim * 4, hipMemcpyHostToDevice));
        ST_TRY(launch_route(idx, pl, io.q, qc, flags, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(out_leaf_count + q0, pl.lcount, (size_t)qc * 4, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < qc; ++q)
            if (out_leaf_count[q0 + q] > leaf_stride)
                return fail(CUDDLE_ERR_BAD_ARG, "leaf_stride %u too small (query %u reaches %u leaves)", leaf_stride, q0 + q,
                            out_leaf_count[q0 + q]);
        const size_t w = (size_t)std::min(pl.lstride, leaf_stride) * 4;
        HIP_TRY(hipMemcpy2D(out_leaves + (uint64_t)q0 * leaf_stride, (size_t)leaf_stride * 4, pl.leaves, (size_t)pl.lstride * 4, w,
                            qc, hipMemcpyDeviceToHost));
    }
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_route_filter_error(cuddle_index *idx, const float *queries, uint32_t n_queries, uint32_t dim,
                                                   uint32_t probes_per_level, uint32_t level, float *out_max_ratio,
                                                   float *out_max_abs) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !out_max_ratio || !out_max_abs) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    if (level >= idx->n_levels || idx->lvl[level].n == 0)
        return fail(CUDDLE_ERR_UNSUPPORTED, "level %u has no contraction operands (narrower than %u nodes)", level, kGemmMinWidth);
    *out_max_ratio = *out_max_abs = 0.f;
    if (n_queries == 0) return CUDDLE_OK;
    ST_TRY(enter_stream(idx, nullptr));
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, 1, probes_per_level, 1, 0, pl));
    const LevelGemm &g = idx->lvl[level];
    if (g.n > pl.fstride) return fail(CUDDLE_ERR_UNSUPPORTED, "level %u is never a whole frontier with probes_per_level %u", level, probes_per_level);
    HostIO io;
    ST_TRY(carve_io(idx, pl.qc, dim, 1, 0, io, nullptr));
    uint32_t *d_max = nullptr; // [0] = max ratio, [1] = max absolute error, as the bits of non-negative floats
    HIP_TRY(hipMalloc((void **)&d_max, 8));
    cuddle_status st = CUDDLE_OK;
    auto body = [&]() -> cuddle_status {
        HIP_TRY(hipMemset(d_max, 0, 8));
        for (uint32_t q0 = 0; q0 < n_queries; q0 += pl.qc) {
            const uint32_t qc = std::min(pl.qc, n_queries - q0);
            HIP_TRY(hipMemcpy(io.q, queries + (uint64_t)q0 * dim, (size_t)qc * dim * 4, hipMemcpyHostToDevice));
            // the filter a search uses on this level: single-pass half precision where its operands exist
            const bool half = g.table16 && !no_half_env() && idx->half_ok[level];
            launch_level_gemm(idx, pl, g, io.q, qc, nullptr, half);
            hipLaunchKernelGGL(route_filter_error_kernel, dim3((g.n + 255) / 256, qc), dim3(256), 0, nullptr, idx->dev, io.q, g.nodes, g.n,
                               pl.fdist, pl.fstride, pl.qsq, g.max_sqn, d_max, half ? 1u : 0u);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(nullptr));
        }
        uint32_t h[2];
        HIP_TRY(hipMemcpy(h, d_max, 8, hipMemcpyDeviceToHost));
        memcpy(out_max_ratio, &h[0], 4);
        memcpy(out_max_abs, &h[1], 4);
        return CUDDLE_OK;
    };
    st = body();
    (void)hipFree(d_max);
    return st;
}

// ---------------------------------------------------------------------------------------------
// index-build side
// ---------------------------------------------------------------------------------------------
namespace {
struct DevBufs { // device allocations of one call, freed on every exit path
    std::vector<void *> p;
    ~DevBufs() { for (void *q : p) (void)hipFree(q); }
    template <class T> cuddle_status alloc(size_t count, T **out) {
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
        p.push_back(q);
        *out = (T *)q;
        return CUDDLE_OK;
    }
};
} // namespace

extern "C" cuddle_status cuddle_kmeans_assign(const float *d_vectors, uint64_t n, const float *d_centroids, uint32_t k,
                                              uint32_t dim, uint32_t metric, uint32_t *d_assign, float *d_dist,
                                              uint64_t *out_fallbacks, int device_id, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_vectors || !d_centroids || !d_assign || !d_dist) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0 || k == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim and k must be >= 1");
    if (metric > CUDDLE_METRIC_DOT) return fail(CUDDLE_ERR_BAD_ARG, "unknown metric %u", metric);
    ST_TRY(use_device(device_id));
    if (out_fallbacks) *out_fallbacks = 0;
    if (n == 0) return CUDDLE_OK;
    const uint32_t kp = split_kp(dim);
    const uint32_t rows_max = (uint32_t)std::min<uint64_t>(n, 1u << 18); // vectors per pass
    const uint32_t gx = (k + 63) / 64;
    DevBufs bufs;
    float *mu, *csq, *asq;
    uint16_t *table, *asplit;
    uint4 *best;
    uint32_t *d_fb;
    ST_TRY(bufs.alloc((size_t)dim, &mu));
    ST_TRY(bufs.alloc((size_t)k, &csq));
    ST_TRY(bufs.alloc((size_t)k * kp * 2, &table));
    ST_TRY(bufs.alloc((size_t)rows_max * kp * 2, &asplit));
    ST_TRY(bufs.alloc((size_t)rows_max, &asq));
    ST_TRY(bufs.alloc((size_t)1, &d_fb));
    HIP_TRY(hipMemsetAsync(mu, 0, dim * sizeof(float), stream));
    HIP_TRY(hipMemsetAsync(d_fb, 0, sizeof(uint32_t), stream));
    if (metric == CUDDLE_METRIC_L2) { // centre on the mean centroid: smaller norms, tighter error bound
        double *acc;
        ST_TRY(bufs.alloc((size_t)dim, &acc));
        HIP_TRY(hipMemsetAsync(acc, 0, dim * sizeof(double), stream));
        hipLaunchKernelGGL(level_sum_kernel, dim3((dim + 63) / 64, 256), dim3(64), 0, stream, d_centroids, (const uint32_t *)nullptr, k, dim, acc);
        hipLaunchKernelGGL(level_mean_kernel, dim3((dim + 255) / 256), dim3(256), 0, stream, acc, k, dim, mu);
    }
    hipLaunchKernelGGL(level_table_kernel, dim3((k + 3) / 4), dim3(256), 0, stream, d_centroids, (const uint32_t *)nullptr, k, dim, kp, mu,
                       table, csq);
    std::vector<float> h(k);
    HIP_TRY(hipMemcpyAsync(h.data(), csq, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float max_csq = 0.f;
    for (float x : h) max_csq = std::max(max_csq, x);
    // column segments of at most kAssignMaxTilesPerSeg tiles, fewer when the grid would not fill the device
    const uint32_t gy_max = (rows_max + kGemmBM - 1) / kGemmBM;
    uint32_t tiles_per_seg = std::min(kAssignMaxTilesPerSeg, gx);
    while (tiles_per_seg > 1 && (uint64_t)gy_max * ((gx + tiles_per_seg - 1) / tiles_per_seg) < 2048) --tiles_per_seg;
    const uint32_t cs = (gx + tiles_per_seg - 1) / tiles_per_seg;
    ST_TRY(bufs.alloc((size_t)rows_max * cs * 2, &best));
    for (uint64_t r0 = 0; r0 < n; r0 += rows_max) {
        const uint32_t rows = (uint32_t)std::min<uint64_t>(rows_max, n - r0);
        const float *vsrc = d_vectors + r0 * dim;
        hipLaunchKernelGGL(center_queries_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, vsrc, mu, rows, dim, kp, asplit, asq);
        const uint32_t gy = (rows + kGemmBM - 1) / kGemmBM;
        if (kp >= 512)
            hipLaunchKernelGGL(assign_gemm_kernel<64>, dim3(gy * cs), dim3(256), assign_gemm_lds_bytes<64>(), stream, (const u32x4 *)asplit,
                               asq, (const u32x4 *)table, csq, rows, k, kp, metric, best, cs, tiles_per_seg);
        else
            hipLaunchKernelGGL(assign_gemm_kernel<32>, dim3(gy * cs), dim3(256), assign_gemm_lds_bytes<32>(), stream, (const u32x4 *)asplit,
                               asq, (const u32x4 *)table, csq, rows, k, kp, metric, best, cs, tiles_per_seg);
        hipLaunchKernelGGL(assign_finish_kernel, dim3((rows + 63) / 64), dim3(256), 0, stream, vsrc, d_centroids, rows, k, dim, metric, best,
                           cs, tiles_per_seg, asq, max_csq, d_assign + r0, d_dist + r0, d_fb);
        HIP_TRY(hipGetLastError());
    }
    uint32_t fb = 0;
    HIP_TRY(hipMemcpyAsync(&fb, d_fb, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream)); // the workspace is freed on return
    if (out_fallbacks) *out_fallbacks = fb;
    return CUDDLE_OK;
}

// A launch's threads (blocks x threads per block) must number fewer than 2^32: larger grids are cut short
// without an error. Kernels whose grid grows with the number of stored vectors are launched in slices.
constexpr uint64_t kMaxBlocksPerLaunch = 1ull << 23; // x 256 threads = 2^31

extern "C" cuddle_status cuddle_kmeans_update(const float *d_vectors, uint64_t n, uint32_t dim, const uint32_t *d_assign,
                                              uint32_t k, float *d_centroids, uint32_t *d_counts, int device_id, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_centroids || !d_counts || ((!d_vectors || !d_assign) && n)) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0 || k == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim and k must be >= 1");
    if (dim > 32768) return fail(CUDDLE_ERR_UNSUPPORTED, "dim %u > 32768", dim);
    if ((uint64_t)k * dim > (1ull << 31)) return fail(CUDDLE_ERR_UNSUPPORTED, "k * dim too large");
    ST_TRY(use_device(device_id));
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)k * 4, stream));
    if (n == 0) return CUDDLE_OK;
    const uint64_t kd = (uint64_t)k * dim, chunks = (n + kUpdateChunk - 1) / kUpdateChunk;
    // chunks per pass: their partial sums ([chunks][k][dim] f32) stay within half a GiB (at least one chunk)
    const uint64_t per_pass = std::max<uint64_t>(1, std::min<uint64_t>(chunks, (512ull << 20) / (kd * 4)));
    // clusters per block: what kUpdateAccFloats accumulators hold; one cluster (and more LDS) for longer vectors
    const uint32_t kr = std::max<uint32_t>(1, std::min<uint32_t>(k, kUpdateAccFloats / dim)), ranges = (k + kr - 1) / kr;
    const size_t partial_lds = (size_t)kr * dim * 4 + kUpdateChunk * 4;
    if (partial_lds > (64u << 10))
        HIP_TRY(hipFuncSetAttribute((const void *)kmeans_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)partial_lds));
    const uint32_t threads = dim <= 64 ? 64 : dim <= 128 ? 128 : 256;
    if (ranges > 65535) return fail(CUDDLE_ERR_UNSUPPORTED, "k * dim too large (%u cluster ranges)", ranges);
    DevBufs bufs;
    float *part, *total;
    ST_TRY(bufs.alloc((size_t)(per_pass * kd), &part));
    ST_TRY(bufs.alloc((size_t)kd, &total));
    HIP_TRY(hipMemsetAsync(total, 0, kd * 4, stream));
    for (uint64_t v0 = 0; v0 < n; v0 += kMaxBlocksPerLaunch * 256) {
        const uint64_t nv = std::min<uint64_t>(n - v0, kMaxBlocksPerLaunch * 256);
        hipLaunchKernelGGL(kmeans_count_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_assign + v0, nv, k, d_counts);
    }
    for (uint64_t c0 = 0; c0 < chunks; c0 += per_pass) {
        const uint32_t nc = (uint32_t)std::min<uint64_t>(per_pass, chunks - c0);
        hipLaunchKernelGGL(kmeans_partial_kernel, dim3(nc, ranges), dim3(threads), partial_lds, stream,
                           d_vectors, d_assign, n, dim, k, kr, c0, part);
        hipLaunchKernelGGL(kmeans_fold_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, stream, part, nc, k, dim, total,
                           d_counts, d_centroids, c0 + nc >= chunks ? 1u : 0u);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(stream)); // the workspace is freed on return
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_kmeanspp_init(const float *d_vectors, uint64_t n, uint32_t dim, uint32_t k, uint32_t metric,
                                              uint64_t first, const float *uniforms, float *d_centroids, uint64_t *chosen,
                                              int device_id, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_vectors || !d_centroids || (k > 1 && !uniforms)) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0 || k == 0 || n == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim, k and n must be >= 1");
    if (k > n) return fail(CUDDLE_ERR_BAD_ARG, "k must be between 1 and the number of vectors"); // kmeans.rs:42
    if (first >= n) return fail(CUDDLE_ERR_BAD_ARG, "first index out of range");
    if (metric > CUDDLE_METRIC_DOT) return fail(CUDDLE_ERR_BAD_ARG, "unknown metric %u", metric);
    ST_TRY(use_device(device_id));
    DevBufs bufs;
    float *mind, *w;
    ST_TRY(bufs.alloc((size_t)n, &mind));
    ST_TRY(bufs.alloc((size_t)n, &w));
    float *hw = nullptr; // pinned: the weights come back every round
    HIP_TRY(hipHostMalloc((void **)&hw, n * sizeof(float), hipHostMallocDefault));
    struct Unpin { float *p; ~Unpin() { (void)hipHostFree(p); } } unpin{hw};
    if (chosen) chosen[0] = first;
    HIP_TRY(hipMemcpyAsync(d_centroids, d_vectors + first * dim, (size_t)dim * 4, hipMemcpyDeviceToDevice, stream));
    for (uint32_t j = 1; j < k; ++j) {
        for (uint64_t v0 = 0; v0 < n; v0 += kMaxBlocksPerLaunch * 64) {
            const uint64_t nv = std::min<uint64_t>(n - v0, kMaxBlocksPerLaunch * 64);
            hipLaunchKernelGGL(kmeanspp_round_kernel, dim3((unsigned)((nv + 63) / 64)), dim3(256), 0, stream, d_vectors + v0 * dim, nv, dim,
                               metric, d_centroids + (size_t)(j - 1) * dim, mind + v0, w + v0, j == 1 ? 1u : 0u);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hw, w, n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        // kmeans.rs:67-77: both folds are sequential single-precision chains, which is what decides the draw
        float total = 0.0f;
        for (uint64_t v = 0; v < n; ++v) total += hw[v];
        float target = uniforms[j - 1] * total;
        uint64_t idx = 0;
        for (uint64_t v = 0; v < n; ++v) {
            target -= hw[v];
            if (target <= 0.0f) {
                idx = v;
                break;
            }
        }
        if (chosen) chosen[j] = idx;
        HIP_TRY(hipMemcpyAsync(d_centroids + (size_t)j * dim, d_vectors + idx * dim, (size_t)dim * 4, hipMemcpyDeviceToDevice, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_binary_encode_device(const float *d_vectors, uint64_t n, uint32_t dim, float threshold,
                                                     uint64_t *d_codes, int device_id, void *stream_) {
    if ((!d_vectors || !d_codes) && n) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim must be >= 1");
    ST_TRY(use_device(device_id));
    const uint32_t W = (dim + 63) / 64;
    const uint64_t per = 4 * kMaxBlocksPerLaunch; // (vector, word) pairs per launch: four per block
    for (uint64_t v0 = 0; v0 < n;) {
        const uint64_t nv = std::min<uint64_t>(n - v0, std::max<uint64_t>(1, per / W));
        hipLaunchKernelGGL(binary_quantize_kernel, dim3((unsigned)((nv * W + 3) / 4)), dim3(256), 0, (hipStream_t)stream_,
                           d_vectors + v0 * dim, nv, dim, W, threshold, d_codes + v0 * W);
        v0 += nv;
    }
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_rabitq_encode_device(const float *d_rot_t, uint32_t dim, const float *d_vectors, uint64_t n,
                                                     uint32_t *d_codes, float *d_norm, float *d_ip_oo, int device_id,
                                                     void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_rot_t || ((!d_vectors || !d_codes || !d_norm || !d_ip_oo) && n)) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (dim == 0) return fail(CUDDLE_ERR_BAD_ARG, "dim must be >= 1");
    constexpr int VB = 8;
    if ((size_t)2 * VB * dim * 4 > (144u << 10)) return fail(CUDDLE_ERR_UNSUPPORTED, "dim %u too large for the encoder", dim);
    ST_TRY(use_device(device_id));
    if (n == 0) return CUDDLE_OK;
    DevBufs bufs;
    float *rot;
    ST_TRY(bufs.alloc((size_t)dim * dim, &rot));
    hipLaunchKernelGGL(transpose_square_kernel, dim3((dim + 255) / 256, dim), dim3(256), 0, stream, d_rot_t, rot, dim);
    const size_t lds = (size_t)2 * VB * dim * 4;
    if (lds > (64u << 10))
        HIP_TRY(hipFuncSetAttribute((const void *)rabitq_encode_kernel<VB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t W32 = (dim + 31) / 32;
    for (uint64_t v0 = 0; v0 < n; v0 += kMaxBlocksPerLaunch * VB) {
        const uint64_t nv = std::min<uint64_t>(n - v0, kMaxBlocksPerLaunch * VB);
        hipLaunchKernelGGL(rabitq_encode_kernel<VB>, dim3((unsigned)((nv + VB - 1) / VB)), dim3(256), lds, stream, d_vectors + v0 * dim, rot,
                           nv, dim, d_codes + v0 * W32, d_norm + v0, d_ip_oo + v0);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream)); // the transposed rotation is freed on return
    return CUDDLE_OK;
}

// Uploads caller-given leaf lists of one chunk into the plan's leaf buffers.
static cuddle_status upload_leaves(cuddle_index *ix, const Plan &pl, const uint32_t *leaves, uint32_t leaf_stride,
                                   const uint32_t *leaf_count, uint32_t q0, uint32_t qc) {
    for (uint32_t q = 0; q < qc; ++q) {
        if (leaf_count[q0 + q] > leaf_stride) return fail(CUDDLE_ERR_BAD_ARG, "leaf_count[%u] > leaf_stride", q0 + q);
        for (uint32_t i = 0; i < leaf_count[q0 + q]; ++i)
            if (leaves[(uint64_t)(q0 + q) * leaf_stride + i] >= ix->dev.n_nodes)
                return fail(CUDDLE_ERR_BAD_ARG, "leaf id out of range (query %u)", q0 + q);
    }
    const size_t w = (size_t)std::min(pl.lstride, leaf_stride) * 4;
    HIP_TRY(hipMemcpy2D(pl.leaves, (size_t)pl.lstride * 4, leaves + (uint64_t)q0 * leaf_stride, (size_t)leaf_stride * 4, w, qc,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl.lcount, leaf_count + q0, (size_t)qc * 4, hipMemcpyHostToDevice));
    return CUDDLE_OK;
}

static uint32_t max_u32(const uint32_t *v, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) m = std::max(m, v[i]);
    return m;
}

extern "C" cuddle_status cuddle_scan_batch(cuddle_index *idx, const float *queries, uint32_t n_queries, uint32_t dim,
                                           const uint32_t *leaves, uint32_t leaf_stride, const uint32_t *leaf_count,
                                           uint64_t *out_pos, uint32_t *out_key, uint32_t cand_stride,
                                           uint32_t *out_cand_count) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !leaves || !leaf_count || !out_pos || !out_key || !out_cand_count) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (n_queries == 0) return CUDDLE_OK;
    ST_TRY(enter_stream(idx, nullptr));
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, 1, 1, 1, max_u32(leaf_count, n_queries), pl));
    HostIO io;
    char *extra = nullptr;
    const size_t per_q = (size_t)cand_stride * 12;
    ST_TRY(carve_io(idx, pl.qc, dim, 1, (size_t)pl.qc * per_q + 512, io, &extra));
    uint64_t *d_pos = (uint64_t *)extra;
    uint32_t *d_key = (uint32_t *)(extra + align_up((size_t)pl.qc * cand_stride * 8, 256));
    idx->timed = false;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += pl.qc) {
        const uint32_t qc = std::min(pl.qc, n_queries - q0);
        HIP_TRY(hipMemcpy(io.q, queries + (uint64_t)q0 * dim, (size_t)qc * dim * 4, hipMemcpyHostToDevice));
        ST_TRY(upload_leaves(idx, pl, leaves, leaf_stride, leaf_count, q0, qc));
        ST_TRY(launch_quantize(idx, pl, io.q, qc, nullptr));
        if (idx->dev.code_kind == CUDDLE_CODE_BINARY) {
            size_t lds = scan_lds_bytes(idx->dev.dim, idx->dev.words, pl.lstride, 0, 0);
            hipLaunchKernelGGL(scan_all_hamming_kernel, dim3(qc), dim3(256), lds, nullptr, idx->dev, pl.qcodes, pl.leaves,
                               pl.lstride, pl.lcount, pl.lstride, d_pos, d_key, cand_stride, pl.cand_count);
        } else {
            const float *rq = reinterpret_cast<const float *>(pl.qcodes), *qn = rq + (size_t)pl.qc * idx->dev.dim;
            size_t lds = rabitq_scan_lds_bytes(idx->dev.dim, pl.lstride, 0, 0);
            hipLaunchKernelGGL(scan_all_rabitq_kernel, dim3(qc), dim3(256), lds, nullptr, idx->dev, rq, qn, pl.leaves,
                               pl.lstride, pl.lcount, pl.lstride, d_pos, d_key, cand_stride, pl.cand_count);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(out_cand_count + q0, pl.cand_count, (size_t)qc * 4, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < qc; ++q)
            if (out_cand_count[q0 + q] > cand_stride)
                return fail(CUDDLE_ERR_BAD_ARG, "cand_stride %u too small (query %u has %u candidates)", cand_stride, q0 + q,
                            out_cand_count[q0 + q]);
        HIP_TRY(hipMemcpy(out_pos + (uint64_t)q0 * cand_stride, d_pos, (size_t)qc * cand_stride * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_key + (uint64_t)q0 * cand_stride, d_key, (size_t)qc * cand_stride * 4, hipMemcpyDeviceToHost));
    }
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_search_leaves_batch(cuddle_index *idx, const float *queries, uint32_t n_queries,
                                                    uint32_t dim, const uint32_t *leaves, uint32_t leaf_stride,
                                                    const uint32_t *leaf_count, uint32_t k, uint32_t rerank_factor,
                                                    uint32_t flags, uint64_t *out_ids, float *out_dist,
                                                    uint32_t *out_count, uint64_t *sel_pos, uint32_t *sel_key,
                                                    float *sel_dist, uint32_t sel_stride, uint32_t *sel_count) {
    if (!idx) return fail(CUDDLE_ERR_BAD_ARG, "index is null");
    if (!queries || !leaves || !leaf_count || !out_ids || !out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    uint64_t rerank_k;
    ST_TRY(rerank_k_of(k, rerank_factor, flags, rerank_k));
    if (n_queries == 0) return CUDDLE_OK;
    ST_TRY(enter_stream(idx, nullptr));
    Plan pl;
    ST_TRY(make_plan(idx, n_queries, k, 1, rerank_k, max_u32(leaf_count, n_queries), pl));
    const bool want_sel = sel_pos && sel_key && sel_count;
    HostIO io;
    char *extra = nullptr;
    const size_t sel_bytes = want_sel ? (size_t)pl.qc * sel_stride * 16 + 1024 : 0;
    ST_TRY(carve_io(idx, pl.qc, dim, k, sel_bytes, io, &extra));
    uint64_t *d_spos = (uint64_t *)extra;
    uint32_t *d_skey = want_sel ? (uint32_t *)(extra + align_up((size_t)pl.qc * sel_stride * 8, 256)) : nullptr;
    float *d_sdist = want_sel ? (float *)((char *)d_skey + align_up((size_t)pl.qc * sel_stride * 4, 256)) : nullptr;
    idx->timed = false;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += pl.qc) {
        const uint32_t qc = std::min(pl.qc, n_queries - q0);
        HIP_TRY(hipMemcpy(io.q, queries + (uint64_t)q0 * dim, (size_t)qc * dim * 4, hipMemcpyHostToDevice));
        ST_TRY(upload_leaves(idx, pl, leaves, leaf_stride, leaf_count, q0, qc));
        ST_TRY(launch_quantize(idx, pl, io.q, qc, nullptr));
        ST_TRY(launch_scan_select(idx, pl, qc, nullptr));
        ST_TRY(launch_rerank(idx, pl, io.q, qc, nullptr));
        ST_TRY(launch_topk(idx, pl, qc, io.ids, io.dist, io.count, nullptr));
        if (want_sel) {
            size_t lds = (size_t)host_next_pow2(pl.sstride) * 8;
            hipLaunchKernelGGL(selection_order_kernel, dim3(qc), dim3(256), lds, nullptr, pl.sel_pos, pl.sel_key, pl.sel_dist,
                               pl.sstride, pl.sel_count, idx->dev.code_kind == CUDDLE_CODE_RABITQ ? 1u : 0u, d_spos, d_skey,
                               d_sdist, sel_stride);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(out_ids + (uint64_t)q0 * k, io.ids, (size_t)qc * k * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_dist + (uint64_t)q0 * k, io.dist, (size_t)qc * k * 4, hipMemcpyDeviceToHost));
        if (out_count) HIP_TRY(hipMemcpy(out_count + q0, io.count, (size_t)qc * 4, hipMemcpyDeviceToHost));
        if (want_sel) {
            HIP_TRY(hipMemcpy(sel_count + q0, pl.sel_count, (size_t)qc * 4, hipMemcpyDeviceToHost));
            for (uint32_t q = 0; q < qc; ++q)
                if (sel_count[q0 + q] > sel_stride) return fail(CUDDLE_ERR_BAD_ARG, "sel_stride %u too small", sel_stride);
            HIP_TRY(hipMemcpy(sel_pos + (uint64_t)q0 * sel_stride, d_spos, (size_t)qc * sel_stride * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(sel_key + (uint64_t)q0 * sel_stride, d_skey, (size_t)qc * sel_stride * 4, hipMemcpyDeviceToHost));
            if (sel_dist)
                HIP_TRY(hipMemcpy(sel_dist + (uint64_t)q0 * sel_stride, d_sdist, (size_t)qc * sel_stride * 4, hipMemcpyDeviceToHost));
        }
    }
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// leaf-sharded exact search: cuddle_search_batch_device split where the shards must agree
// ---------------------------------------------------------------------------------------------
// Leaf-list capacity a plan for this probes_per_level uses (make_plan's lstride; it does not depend on
// the batch size or k).
static uint32_t plan_leaf_stride(const cuddle_index *ix, uint32_t probes) {
    probes = std::min(probes, kSortCap);
    uint64_t fb = ix->dev.n_roots, lsum = 0;
    for (uint32_t l = 0; l < ix->n_levels; ++l) {
        fb = std::min<uint64_t>(fb, ix->level_width[l]);
        const uint64_t sel = std::min<uint64_t>(fb, probes);
        lsum += std::min<uint64_t>(sel, ix->level_leaves[l]);
        fb = sel * std::max<uint32_t>(ix->max_children, 1);
    }
    return (uint32_t)align_up(std::max<uint64_t>(std::min<uint64_t>(lsum, ix->n_leaves), 1), 4);
}

extern "C" cuddle_status cuddle_index_leaf_stride(const cuddle_index *idx, uint32_t probes_per_level, uint32_t *out) {
    if (!idx || !out) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    *out = plan_leaf_stride(idx, probes_per_level);
    return CUDDLE_OK;
}

// Caller-given leaf lists -> the plan's: out-of-range node ids are clamped (a non-leaf id scans nothing).
__global__ void adopt_leaves_kernel(const uint32_t *src, uint32_t src_stride, const uint32_t *src_count, uint32_t nq,
                                    uint32_t n_nodes, uint32_t *dst, uint32_t dst_stride, uint32_t *dst_count) {
    const uint32_t q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t n = min(src_count[q], min(src_stride, dst_stride));
    if (i < n) dst[(uint64_t)q * dst_stride + i] = min(src[(uint64_t)q * src_stride + i], n_nodes - 1);
    if (i == 0) dst_count[q] = n;
}

extern "C" cuddle_status cuddle_route_batch_device(cuddle_index *idx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                                                   uint32_t probes_per_level, uint32_t flags, uint32_t *d_leaves,
                                                   uint32_t leaf_stride, uint32_t *d_leaf_count, void *stream) {
    if (!idx || !d_queries || !d_leaves || !d_leaf_count) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    if (n_queries == 0) return CUDDLE_OK;
    if (n_queries > kMaxChunk) return fail(CUDDLE_ERR_UNSUPPORTED, "1..%u queries per call", kMaxChunk);
    if (leaf_stride < plan_leaf_stride(idx, probes_per_level))
        return fail(CUDDLE_ERR_BAD_ARG, "leaf_stride %u < cuddle_index_leaf_stride() = %u", leaf_stride, plan_leaf_stride(idx, probes_per_level));
    hipStream_t s = static_cast<hipStream_t>(stream);
    idx->shard_state = 0;
    idx->timed = (flags & CUDDLE_SEARCH_TIME_STAGES) != 0 && idx->events_used + 4096 < kMaxEvents;
    ST_TRY(enter_stream(idx, s));
    Plan &pl = idx->shard_plan; // held by the handle: the copies below are asynchronous
    pl = Plan();
    ST_TRY(make_plan(idx, n_queries, 1, probes_per_level, 1, 0, pl));
    if (pl.qc < n_queries) return fail(CUDDLE_ERR_UNSUPPORTED, "%u queries do not fit the work-space budget in one chunk (%u do)", n_queries, pl.qc);
    mark(idx, s, CUDDLE_STAGE_ROUTE);
    ST_TRY(launch_route(idx, pl, d_queries, n_queries, flags, s));
    HIP_TRY(hipMemcpy2DAsync(d_leaves, (size_t)leaf_stride * 4, pl.leaves, (size_t)pl.lstride * 4, (size_t)pl.lstride * 4, n_queries,
                             hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_leaf_count, pl.lcount, (size_t)n_queries * 4, hipMemcpyDeviceToDevice, s));
    mark(idx, s, -1);
    return CUDDLE_OK;
}

// d_hist != null: binary codes, histogram mode. d_keys != null: RaBitQ codes, the rank's own selection and its keys.
static cuddle_status shard_scan_impl(cuddle_index *idx, const float *d_queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                     uint32_t probes_per_level, uint32_t rerank_factor, uint32_t flags, const uint32_t *d_leaves,
                                     uint32_t leaf_stride_in, const uint32_t *d_leaf_count, uint32_t *d_hist,
                                     uint32_t *leaf_stride, void *stream, uint64_t *d_keys = nullptr) {
    if (!idx || !d_queries || (!d_hist && !d_keys)) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    idx->shard_state = 0;
    if (d_hist && idx->dev.code_kind != CUDDLE_CODE_BINARY)
        return fail(CUDDLE_ERR_UNSUPPORTED, "cuddle_shard_scan is for binary codes (RaBitQ codes: cuddle_shard_scan_rabitq)");
    if (d_keys && idx->dev.code_kind != CUDDLE_CODE_RABITQ)
        return fail(CUDDLE_ERR_UNSUPPORTED, "cuddle_shard_scan_rabitq is for RaBitQ codes (binary codes: cuddle_shard_scan)");
    if (dim != idx->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, idx->dev.dim);
    if (probes_per_level == 0) return fail(CUDDLE_ERR_BAD_ARG, "probes_per_level must be >= 1");
    if (n_queries == 0 || n_queries > kMaxChunk) return fail(CUDDLE_ERR_UNSUPPORTED, "1..%u queries per sharded batch", kMaxChunk);
    uint64_t rerank_k;
    ST_TRY(rerank_k_of(k, rerank_factor, flags, rerank_k));
    hipStream_t s = static_cast<hipStream_t>(stream);
    idx->timed = (flags & CUDDLE_SEARCH_TIME_STAGES) != 0 && idx->events_used + 4096 < kMaxEvents;
    ST_TRY(enter_stream(idx, s));
    idx->last_levels = idx->n_levels;
    HIP_TRY(hipMemsetAsync(idx->d_stats, 0, sizeof(DevStats), s));
    Plan &pl = idx->shard_plan;
    pl = Plan();
    ST_TRY(make_plan(idx, n_queries, k, probes_per_level, rerank_k, 0, pl, nullptr, 0, /*whole_index=*/false));
    if (pl.qc < n_queries) return fail(CUDDLE_ERR_UNSUPPORTED, "%u queries do not fit the work-space budget in one chunk (%u do)", n_queries, pl.qc);
    if (idx->timed) idx->timed_searches++;
    mark(idx, s, CUDDLE_STAGE_QUANTIZE);
    ST_TRY(launch_quantize(idx, pl, d_queries, n_queries, s));
    mark(idx, s, CUDDLE_STAGE_ROUTE);
    if (d_leaves) {
        hipLaunchKernelGGL(adopt_leaves_kernel, dim3((pl.lstride + 255) / 256, n_queries), dim3(256), 0, s, d_leaves, leaf_stride_in,
                           d_leaf_count, n_queries, idx->dev.n_nodes, pl.leaves, pl.lstride, pl.lcount);
        HIP_TRY(hipGetLastError());
    } else {
        ST_TRY(launch_route(idx, pl, d_queries, n_queries, flags, s));
    }
    mark(idx, s, CUDDLE_STAGE_SCAN);
    if (d_keys) {
        if (pl.lstride >= (1u << kRqSlotBits)) return fail(CUDDLE_ERR_UNSUPPORTED, "leaf lists of %u slots: the key holds %u bits", pl.lstride, kRqSlotBits);
        ST_TRY(launch_scan_select(idx, pl, n_queries, s));
        hipLaunchKernelGGL(rq_exchange_keys_kernel, dim3(n_queries), dim3(256), rq_exchange_lds_bytes(pl.lstride), s, idx->dev, pl.leaves, pl.lstride,
                           pl.lcount, pl.lstride, pl.sel_pos, pl.sel_key, pl.sstride, pl.sel_count, pl.rerank_k, d_keys, pl.sel_ord);
        HIP_TRY(hipGetLastError());
    } else {
        ST_TRY(launch_scan_select(idx, pl, n_queries, s, d_hist));
    }
    mark(idx, s, -1);
    idx->shard_nq = n_queries;
    idx->shard_state = d_keys ? 3 : 1;
    if (leaf_stride) *leaf_stride = pl.lstride;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_rerank_k(uint32_t k, uint32_t rerank_factor, uint32_t flags, uint32_t *out) {
    if (!out) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    uint64_t r;
    ST_TRY(rerank_k_of(k, rerank_factor, flags, r));
    *out = (uint32_t)r;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_shard_scan_rabitq(cuddle_index *idx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                                                  uint32_t k, uint32_t probes_per_level, uint32_t rerank_factor, uint32_t flags,
                                                  const uint32_t *d_leaves, uint32_t leaf_stride_in, const uint32_t *d_leaf_count,
                                                  uint64_t *d_keys, void *stream) {
    if (!d_keys || ((d_leaves == nullptr) != (d_leaf_count == nullptr))) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    return shard_scan_impl(idx, d_queries, n_queries, dim, k, probes_per_level, rerank_factor, flags, d_leaves, leaf_stride_in,
                           d_leaf_count, nullptr, nullptr, stream, d_keys);
}

extern "C" cuddle_status cuddle_shard_finish_rabitq(cuddle_index *idx, const float *d_queries, const uint64_t *d_all_keys,
                                                    uint32_t n_shards, uint64_t *d_out_ids, float *d_out_dist,
                                                    uint64_t *d_out_order, uint32_t *d_out_count, void *stream) {
    if (!idx || !d_queries || !d_all_keys || !d_out_ids || !d_out_dist || !d_out_order || n_shards == 0)
        return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (idx->shard_state != 3) return fail(CUDDLE_ERR_BAD_ARG, "cuddle_shard_finish_rabitq must follow cuddle_shard_scan_rabitq");
    const Plan &pl = idx->shard_plan;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nq = idx->shard_nq;
    if ((uint64_t)n_shards * pl.rerank_k > 0xFFFFFFFFull) return fail(CUDDLE_ERR_UNSUPPORTED, "too many keys per query");
    ST_TRY(enter_stream(idx, s));
    mark(idx, s, CUDDLE_STAGE_SCAN);
    uint64_t *thr = reinterpret_cast<uint64_t *>(pl.thr); // [nq] u64
    hipLaunchKernelGGL(rq_threshold_kernel, dim3(nq), dim3(256), 0, s, d_all_keys, n_shards, nq, pl.rerank_k, thr);
    hipLaunchKernelGGL(rq_cut_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, pl.sel_ord, pl.sstride, pl.sel_count, pl.rerank_k, thr, nq);
    HIP_TRY(hipGetLastError());
    mark(idx, s, CUDDLE_STAGE_RERANK);
    ST_TRY(launch_rerank(idx, pl, d_queries, nq, s, false));
    mark(idx, s, CUDDLE_STAGE_TOPK);
    ST_TRY(launch_topk(idx, pl, nq, d_out_ids, d_out_dist, d_out_count, s, d_out_order));
    mark(idx, s, -1);
    idx->shard_state = 0;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_shard_scan(cuddle_index *idx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                                           uint32_t k, uint32_t probes_per_level, uint32_t rerank_factor, uint32_t flags,
                                           uint32_t *d_hist, uint32_t *leaf_stride, void *stream) {
    return shard_scan_impl(idx, d_queries, n_queries, dim, k, probes_per_level, rerank_factor, flags, nullptr, 0, nullptr, d_hist,
                           leaf_stride, stream);
}

extern "C" cuddle_status cuddle_shard_scan_leaves(cuddle_index *idx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                                                  uint32_t k, uint32_t probes_per_level, uint32_t rerank_factor, uint32_t flags,
                                                  const uint32_t *d_leaves, uint32_t leaf_stride_in, const uint32_t *d_leaf_count,
                                                  uint32_t *d_hist, uint32_t *leaf_stride, void *stream) {
    if (!d_leaves || !d_leaf_count) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    return shard_scan_impl(idx, d_queries, n_queries, dim, k, probes_per_level, rerank_factor, flags, d_leaves, leaf_stride_in,
                           d_leaf_count, d_hist, leaf_stride, stream);
}

extern "C" cuddle_status cuddle_shard_ties(cuddle_index *idx, const uint32_t *d_global_hist, uint32_t *d_tie_count,
                                           void *stream) {
    if (!idx || !d_global_hist || !d_tie_count) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (idx->shard_state != 1) return fail(CUDDLE_ERR_BAD_ARG, "cuddle_shard_ties must follow cuddle_shard_scan");
    const Plan &pl = idx->shard_plan;
    const IndexDev &v = idx->dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ST_TRY(enter_stream(idx, s));
    const size_t lds = scan_lds_bytes(v.dim, v.words, pl.lstride, 0, 0);
    mark(idx, s, CUDDLE_STAGE_SCAN); // the selection steps count as scan time
    if (v.dim < 255)
        hipLaunchKernelGGL(shard_ties_kernel<uint8_t>, dim3(idx->shard_nq), dim3(256), lds, s, v, pl.leaves, pl.lstride, pl.lcount,
                           pl.lstride, d_global_hist, pl.rerank_k, (const void *)pl.gdist, pl.cstride, pl.thr, d_tie_count);
    else
        hipLaunchKernelGGL(shard_ties_kernel<uint16_t>, dim3(idx->shard_nq), dim3(256), lds, s, v, pl.leaves, pl.lstride, pl.lcount,
                           pl.lstride, d_global_hist, pl.rerank_k, (const void *)pl.gdist, pl.cstride, pl.thr, d_tie_count);
    HIP_TRY(hipGetLastError());
    mark(idx, s, -1);
    idx->shard_state = 2;
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_shard_finish(cuddle_index *idx, const float *d_queries, const uint32_t *d_global_ties,
                                             uint64_t *d_out_ids, float *d_out_dist, uint64_t *d_out_order,
                                             uint32_t *d_out_count, void *stream) {
    if (!idx || !d_queries || !d_global_ties || !d_out_ids || !d_out_dist || !d_out_order)
        return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    ST_TRY(use_device(idx->device));
    if (idx->shard_state != 2) return fail(CUDDLE_ERR_BAD_ARG, "cuddle_shard_finish must follow cuddle_shard_ties");
    const Plan &pl = idx->shard_plan;
    const IndexDev &v = idx->dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nq = idx->shard_nq;
    ST_TRY(enter_stream(idx, s));
    const size_t lds = scan_lds_bytes(v.dim, v.words, pl.lstride, 0, 0);
    mark(idx, s, CUDDLE_STAGE_SCAN);
    if (v.dim < 255)
        hipLaunchKernelGGL(shard_select_kernel<uint8_t>, dim3(nq), dim3(256), lds, s, v, pl.leaves, pl.lstride, pl.lcount,
                           pl.lstride, pl.thr, d_global_ties, (const void *)pl.gdist, pl.cstride, pl.sel_pos, pl.sel_key,
                           pl.sel_ord, pl.sstride, pl.sel_count);
    else
        hipLaunchKernelGGL(shard_select_kernel<uint16_t>, dim3(nq), dim3(256), lds, s, v, pl.leaves, pl.lstride, pl.lcount,
                           pl.lstride, pl.thr, d_global_ties, (const void *)pl.gdist, pl.cstride, pl.sel_pos, pl.sel_key,
                           pl.sel_ord, pl.sstride, pl.sel_count);
    HIP_TRY(hipGetLastError());
    mark(idx, s, CUDDLE_STAGE_RERANK);
    ST_TRY(launch_rerank(idx, pl, d_queries, nq, s, false));
    mark(idx, s, CUDDLE_STAGE_TOPK);
    ST_TRY(launch_topk(idx, pl, nq, d_out_ids, d_out_dist, d_out_count, s, d_out_order));
    mark(idx, s, -1);
    idx->shard_state = 0;
    return CUDDLE_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU merge (host)
// ---------------------------------------------------------------------------------------------
extern "C" cuddle_status cuddle_merge_topk_ordered_device(const uint64_t *d_lists_ids, const float *d_lists_dist,
                                                          const uint64_t *d_lists_order, uint32_t n_lists, uint32_t n_queries,
                                                          uint32_t k, uint64_t list_stride_bytes, uint64_t *d_out_ids,
                                                          float *d_out_dist, uint32_t *d_out_count, void *stream) {
    if (!d_lists_ids || !d_lists_dist || !d_lists_order || !d_out_ids || !d_out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (k == 0 || n_lists == 0) return fail(CUDDLE_ERR_BAD_ARG, "k and n_lists must be >= 1");
    if ((uint64_t)n_lists * k > 64 * 16) return fail(CUDDLE_ERR_UNSUPPORTED, "device merge handles n_lists * k <= 1024");
    if (list_stride_bytes % 8) return fail(CUDDLE_ERR_BAD_ARG, "list_stride_bytes must be a multiple of 8");
    if (n_queries == 0) return CUDDLE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t total = n_lists * k;
    const dim3 grid((n_queries + 3) / 4), block(256);
    const uint64_t dense = (uint64_t)n_queries * k; // list_stride_bytes = 0: each array is [n_lists][n_queries][k]
    const uint64_t si = list_stride_bytes ? list_stride_bytes : dense * 8, sd = list_stride_bytes ? list_stride_bytes : dense * 4,
                   so = si;
#define CUDDLE_MERGE(KPL)                                                                                              \
    hipLaunchKernelGGL(merge_ordered_kernel<KPL>, grid, block, 0, s, d_lists_ids, d_lists_dist, d_lists_order, si, sd, \
                       so, n_lists, n_queries, k, d_out_ids, d_out_dist, d_out_count)
    if (total <= 64) CUDDLE_MERGE(1);
    else if (total <= 128) CUDDLE_MERGE(2);
    else if (total <= 256) CUDDLE_MERGE(4);
    else if (total <= 512) CUDDLE_MERGE(8);
    else CUDDLE_MERGE(16);
#undef CUDDLE_MERGE
    HIP_TRY(hipGetLastError());
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_merge_topk_ordered(const uint64_t *lists_ids, const float *lists_dist,
                                                   const uint64_t *lists_order, uint32_t n_lists, uint32_t n_queries,
                                                   uint32_t k, uint64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (!lists_ids || !lists_dist || !lists_order || !out_ids || !out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (k == 0 || n_lists == 0) return fail(CUDDLE_ERR_BAD_ARG, "k and n_lists must be >= 1");
    struct Item { float d; uint64_t order, id; };
    std::vector<Item> items;
    items.reserve((size_t)n_lists * k);
    for (uint32_t q = 0; q < n_queries; ++q) {
        items.clear();
        for (uint32_t l = 0; l < n_lists; ++l)
            for (uint32_t r = 0; r < k; ++r) {
                size_t o = ((size_t)l * n_queries + q) * k + r;
                if (lists_ids[o] == UINT64_MAX) break;
                items.push_back({lists_dist[o], lists_order[o], lists_ids[o]});
            }
        std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
            if (a.d < b.d) return true;
            if (a.d > b.d) return false;
            return a.order < b.order;
        });
        uint32_t n = (uint32_t)std::min<size_t>(items.size(), k);
        for (uint32_t i = 0; i < k; ++i) {
            out_ids[(size_t)q * k + i] = i < n ? items[i].id : UINT64_MAX;
            out_dist[(size_t)q * k + i] = i < n ? items[i].d : std::numeric_limits<float>::infinity();
        }
        if (out_count) out_count[q] = n;
    }
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_merge_topk(const uint64_t *lists_ids, const float *lists_dist, uint32_t n_lists,
                                           uint32_t n_queries, uint32_t k, uint64_t *out_ids, float *out_dist,
                                           uint32_t *out_count) {
    if (!lists_ids || !lists_dist || !out_ids || !out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null pointer");
    if (k == 0 || n_lists == 0) return fail(CUDDLE_ERR_BAD_ARG, "k and n_lists must be >= 1");
    struct Item { float d; uint32_t list, rank; uint64_t id; };
    std::vector<Item> items;
    items.reserve((size_t)n_lists * k);
    for (uint32_t q = 0; q < n_queries; ++q) {
        items.clear();
        for (uint32_t l = 0; l < n_lists; ++l)
            for (uint32_t r = 0; r < k; ++r) {
                size_t o = ((size_t)l * n_queries + q) * k + r;
                if (lists_ids[o] == UINT64_MAX) break;
                items.push_back({lists_dist[o], l, r, lists_ids[o]});
            }
        std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
            if (a.d < b.d) return true;
            if (a.d > b.d) return false;
            if (a.list != b.list) return a.list < b.list;
            return a.rank < b.rank;
        });
        uint32_t n = (uint32_t)std::min<size_t>(items.size(), k);
        for (uint32_t i = 0; i < k; ++i) {
            out_ids[(size_t)q * k + i] = i < n ? items[i].id : UINT64_MAX;
            out_dist[(size_t)q * k + i] = i < n ? items[i].d : std::numeric_limits<float>::infinity();
        }
        if (out_count) out_count[q] = n;
    }
    return CUDDLE_OK;
}

#include "cuddle_comm.hpp"
