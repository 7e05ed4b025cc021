// cuddle_comm.hpp — leaf-sharded search over the GPUs of one node behind the C ABI: one process per
// GPU, RCCL collectives over xGMI. Included at the end of cuddle_hip.hip (host code only; it uses the
// handle and the cuddle_shard_* steps defined there).
//
// The reference is a single process (sharding is prose only, docs/goals/SCALING_GOALS.md:139-147).
// Protocol per batch (see cuddle_shard.hpp for why it is exact): every rank routes 1/world of the
// queries, the leaf lists are ALL-GATHERed; every rank scans its own leaves for the whole batch and
// the per-query Hamming histograms are ALL-REDUCEd; the per-leaf-slot tie counts at the threshold
// are ALL-REDUCEd; every rank reranks its members of the global selection and the local top-k
// (id, order key, distance) are ALL-GATHERed and merged by (distance, order) on every rank.
// All messages are a few MB at most: latency-bound on the fully connected xGMI links.
//
// RCCL is loaded with dlopen the first time a communicator is made, so that the library has no link
// dependency on it (hosts that never shard never load it; a host that already carries an RCCL -
// e.g. a process that imported torch - shares that copy).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void *handle = nullptr;
    std::string error;
};

RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("CUDDLE_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            api.error = dlerror();
        }
        if (!api.handle) return;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(api.handle, name);
            if (!p) api.error = std::string("missing RCCL symbol ") + name;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather || !api.GetErrorString) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api;
}

cuddle_status rccl_ready() {
    RcclApi &a = rccl_api();
    if (!a.handle) return fail(CUDDLE_ERR_RCCL, "RCCL is not available: %s", a.error.empty() ? "librccl.so.1 not found" : a.error.c_str());
    return CUDDLE_OK;
}

#define RCCL_TRY(expr)                                                                                             \
    do {                                                                                                           \
        ncclResult_t r__ = (expr);                                                                                 \
        if (r__ != ncclSuccess)                                                                                    \
            return fail(CUDDLE_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, rccl_api().GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

} // namespace

struct cuddle_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    DevBuf buf; // messages of one batch
    std::mutex mu;
    // Before the first data collective of a batch shape the ranks agree that every one of them can run it (see
    // sharded_preflight): the agreement is remembered for the shape it was made for.
    uint32_t *d_agree = nullptr; // [4] device words of the agreement all-reduce
    uint64_t agreed_shape = 0;
    const cuddle_index *agreed_shard = nullptr;
};

static_assert(CUDDLE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as an opaque byte array of RCCL's size");

extern "C" cuddle_status cuddle_comm_unique_id(uint8_t *id) {
    if (!id) return fail(CUDDLE_ERR_BAD_ARG, "id is null");
    ST_TRY(rccl_ready());
    ncclUniqueId u;
    RCCL_TRY(rccl_api().GetUniqueId(&u));
    memcpy(id, u.internal, CUDDLE_COMM_ID_BYTES);
    return CUDDLE_OK;
}

extern "C" cuddle_status cuddle_comm_init(int rank, int world, const uint8_t *id, int device_id, cuddle_comm **out) {
    if (!out) return fail(CUDDLE_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(CUDDLE_ERR_BAD_ARG, "bad rank %d / world %d / id", rank, world);
    ST_TRY(rccl_ready());
    ST_TRY(use_device(device_id));
    cuddle_comm *c = new (std::nothrow) cuddle_comm();
    if (!c) return fail(CUDDLE_ERR_OOM, "host allocation failed");
    c->rank = rank; c->world = world; c->device = device_id;
    ncclUniqueId u;
    memcpy(u.internal, id, CUDDLE_COMM_ID_BYTES);
    ncclResult_t r = rccl_api().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(CUDDLE_ERR_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_api().GetErrorString(r));
    }
    if (hipMalloc((void **)&c->d_agree, 16) != hipSuccess) {
        (void)hipGetLastError();
        (void)rccl_api().CommDestroy(c->comm);
        delete c;
        return fail(CUDDLE_ERR_OOM, "hipMalloc(16) failed");
    }
    *out = c;
    return CUDDLE_OK;
}

extern "C" void cuddle_comm_destroy(cuddle_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm) (void)rccl_api().CommDestroy(c->comm);
    if (c->d_agree) (void)hipFree(c->d_agree);
    c->buf.release();
    delete c;
}

extern "C" int cuddle_comm_rank(const cuddle_comm *c) { return c ? c->rank : -1; }
extern "C" int cuddle_comm_world(const cuddle_comm *c) { return c ? c->world : 0; }

// Everything about a batch shape that can fail on ONE rank only - limits that depend on the rank's own shard (its
// leaf sizes decide whether the chunk fits the work-space budget, the candidate bound, the scan's LDS), growing the
// work space or the message buffer - is tried here, before any data travels, and the ranks then agree on the outcome
// with one small all-reduce: a rank that left the protocol on its own would leave its peers waiting in the next
// collective for good. Also checked: every rank uses the same leaf-list stride (the messages are laid out by it).
// The agreement costs a host wait, so it is made once per (shard, batch shape) and remembered.
static cuddle_status sharded_preflight(cuddle_index *shard, cuddle_comm *comm, uint32_t nq, uint32_t k, uint32_t probes,
                                       uint32_t rerank_factor, uint32_t flags, uint32_t lstride, size_t msg_bytes, hipStream_t s) {
    const uint64_t shape = ((uint64_t)nq << 40) ^ ((uint64_t)k << 28) ^ ((uint64_t)probes << 14) ^ ((uint64_t)rerank_factor << 3) ^ flags;
    if (comm->agreed_shard == shard && comm->agreed_shape == shape && msg_bytes <= comm->buf.bytes) return CUDDLE_OK;
    RcclApi &api = rccl_api();
    cuddle_status local = CUDDLE_OK;
    std::string local_msg;
    {
        uint64_t rerank_k = 0;
        local = rerank_k_of(k, rerank_factor, flags, rerank_k);
        if (local == CUDDLE_OK) {
            std::lock_guard<std::mutex> lk(shard->mu);
            Plan route_plan, scan_plan;
            local = make_plan(shard, nq, 1, probes, 1, 0, route_plan);
            if (local == CUDDLE_OK && route_plan.qc < nq)
                local = fail(CUDDLE_ERR_UNSUPPORTED, "%u queries do not fit the work-space budget in one chunk (%u do)", nq, route_plan.qc);
            if (local == CUDDLE_OK) local = make_plan(shard, nq, k, probes, rerank_k, 0, scan_plan, nullptr, 0, /*whole_index=*/false);
            if (local == CUDDLE_OK && scan_plan.qc < nq)
                local = fail(CUDDLE_ERR_UNSUPPORTED, "%u queries do not fit the work-space budget in one chunk (%u do)", nq, scan_plan.qc);
            if (local == CUDDLE_OK && shard->dev.code_kind == CUDDLE_CODE_RABITQ && scan_plan.lstride >= (1u << kRqSlotBits))
                local = fail(CUDDLE_ERR_UNSUPPORTED, "leaf lists of %u slots: the key holds %u bits", scan_plan.lstride, kRqSlotBits);
        }
        if (local == CUDDLE_OK) local = comm->buf.ensure(msg_bytes);
        if (local != CUDDLE_OK) local_msg = g_last_error;
    }
    // max over the ranks of (status, stride, ~stride): any failure shows, and so do two different strides
    const uint32_t h[4] = {(uint32_t)local, lstride, ~lstride, 0u};
    uint32_t g[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(comm->d_agree, h, 16, hipMemcpyHostToDevice, s));
    RCCL_TRY(api.AllReduce(comm->d_agree, comm->d_agree, 4, ncclUint32, ncclMax, comm->comm, s));
    HIP_TRY(hipMemcpyAsync(g, comm->d_agree, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (local != CUDDLE_OK) return fail(local, "%s", local_msg.c_str());
    if (g[0] != CUDDLE_OK)
        return fail((cuddle_status)g[0], "another rank cannot run this batch on its shard (status %u there); no rank has started it", g[0]);
    if (g[1] != lstride || ~g[2] != lstride)
        return fail(CUDDLE_ERR_BAD_INDEX, "the ranks' trees differ: leaf-list strides %u..%u for probes_per_level %u", ~g[2], g[1], probes);
    comm->agreed_shard = shard;
    comm->agreed_shape = shape;
    return CUDDLE_OK;
}

// One round of the protocol for at most kMaxChunk queries (the caller holds comm->mu and has made the device current).
static cuddle_status sharded_search_chunk(cuddle_index *shard, cuddle_comm *comm, const float *d_queries, uint32_t n_queries,
                                          uint32_t dim, uint32_t k, uint32_t probes_per_level, uint32_t rerank_factor,
                                          uint32_t flags, uint64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                                          void *stream) {
    RcclApi &api = rccl_api();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t world = (uint32_t)comm->world, rank = (uint32_t)comm->rank, nq = n_queries, D = shard->dev.dim;
    uint32_t lstride = 0;
    ST_TRY(cuddle_index_leaf_stride(shard, probes_per_level, &lstride));
    const uint32_t per = (nq + world - 1) / world; // queries routed per rank
    // message buffers of this batch
    const size_t msg_leaves = (size_t)per * (lstride + 1) * 4;                 // [per][lstride] lists + [per] counts
    const size_t res_words = 2 * (size_t)nq * k + ((size_t)nq * k + 1) / 2;     // ids, order keys, f32 distances (8-byte words)
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const bool rabitq = shard->dev.code_kind == CUDDLE_CODE_RABITQ;
    uint32_t rerank_k = 0;
    ST_TRY(cuddle_rerank_k(k, rerank_factor, flags, &rerank_k));
    const size_t key_words = rabitq ? (size_t)nq * rerank_k : 0; // RaBitQ: one key per member of a rank's own selection
    const size_t o_lmine = carve(msg_leaves), o_lall = carve(msg_leaves * world), o_lv = carve((size_t)world * per * lstride * 4),
                 o_lc = carve((size_t)world * per * 4), o_hist = carve(rabitq ? 0 : (size_t)nq * (D + 1) * 4),
                 o_ties = carve(rabitq ? 0 : (size_t)nq * lstride * 4), o_rmine = carve(res_words * 8), o_rall = carve(res_words * 8 * world),
                 o_kmine = carve(key_words * 8), o_kall = carve(key_words * 8 * world);
    // shard-local checks and allocations first, agreed on by all ranks: nothing below may fail on one rank alone
    ST_TRY(sharded_preflight(shard, comm, nq, k, probes_per_level, rerank_factor, flags, lstride, off, s));
    char *b = static_cast<char *>(comm->buf.p);
    uint32_t *l_mine = (uint32_t *)(b + o_lmine), *l_all = (uint32_t *)(b + o_lall), *lv = (uint32_t *)(b + o_lv), *lc = (uint32_t *)(b + o_lc);
    uint32_t *hist = (uint32_t *)(b + o_hist), *ties = (uint32_t *)(b + o_ties);
    uint64_t *r_mine = (uint64_t *)(b + o_rmine), *r_all = (uint64_t *)(b + o_rall);

    // 1. routing, split by queries: this rank routes queries [q0, q1); lists + counts travel as one message
    const uint32_t q0 = std::min(nq, rank * per), q1 = std::min(nq, q0 + per);
    HIP_TRY(hipMemsetAsync(l_mine, 0, msg_leaves, s));
    if (q1 > q0)
        ST_TRY(cuddle_route_batch_device(shard, d_queries + (uint64_t)q0 * D, q1 - q0, dim, probes_per_level, flags, l_mine, lstride,
                                         l_mine + (size_t)per * lstride, s));
    RCCL_TRY(api.AllGather(l_mine, l_all, msg_leaves / 4, ncclUint32, comm->comm, s));
    // un-interleave the ranks' messages into [world * per][lstride] lists and [world * per] counts
    for (uint32_t r = 0; r < world; ++r) {
        const char *m = (const char *)l_all + (size_t)r * msg_leaves;
        HIP_TRY(hipMemcpyAsync(lv + (size_t)r * per * lstride, m, (size_t)per * lstride * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(lc + (size_t)r * per, m + (size_t)per * lstride * 4, (size_t)per * 4, hipMemcpyDeviceToDevice, s));
    }
    const size_t n = (size_t)nq * k;
    if (rabitq) {
        // 2'. scan + this rank's own selection -> its keys, gathered from every rank; 3'. the ranks agree on the
        // rerank_k-th smallest key, each keeps its selection up to it; 4'. as below
        uint64_t *k_mine = (uint64_t *)(b + o_kmine), *k_all = (uint64_t *)(b + o_kall);
        ST_TRY(cuddle_shard_scan_rabitq(shard, d_queries, nq, dim, k, probes_per_level, rerank_factor, flags, lv, lstride, lc, k_mine, s));
        RCCL_TRY(api.AllGather(k_mine, k_all, key_words, ncclUint64, comm->comm, s));
        ST_TRY(cuddle_shard_finish_rabitq(shard, d_queries, k_all, world, r_mine, (float *)(r_mine + 2 * n), r_mine + n, nullptr, s));
        RCCL_TRY(api.AllGather(r_mine, r_all, res_words, ncclUint64, comm->comm, s));
        return cuddle_merge_topk_ordered_device(r_all, (const float *)(r_all + 2 * n), r_all + n, world, nq, k, res_words * 8, d_out_ids,
                                                d_out_dist, d_out_count, s);
    }
    // 2. scan of this rank's leaves for the whole batch -> histograms summed over the ranks
    uint32_t stride_used = 0;
    ST_TRY(cuddle_shard_scan_leaves(shard, d_queries, nq, dim, k, probes_per_level, rerank_factor, flags, lv, lstride, lc, hist,
                                    &stride_used, s));
    if (stride_used != lstride) return fail(CUDDLE_ERR_HIP, "leaf stride changed between the routing and the scan");
    RCCL_TRY(api.AllReduce(hist, hist, (size_t)nq * (D + 1), ncclUint32, ncclSum, comm->comm, s));
    // 3. ties at the threshold per leaf slot -> summed over the ranks
    ST_TRY(cuddle_shard_ties(shard, hist, ties, s));
    RCCL_TRY(api.AllReduce(ties, ties, (size_t)nq * lstride, ncclUint32, ncclSum, comm->comm, s));
    // 4. rerank + local top-k, written as one message: ids | order keys | distances
    ST_TRY(cuddle_shard_finish(shard, d_queries, ties, r_mine, (float *)(r_mine + 2 * n), r_mine + n, nullptr, s));
    RCCL_TRY(api.AllGather(r_mine, r_all, res_words, ncclUint64, comm->comm, s));
    return cuddle_merge_topk_ordered_device(r_all, (const float *)(r_all + 2 * n), r_all + n, world, nq, k, res_words * 8, d_out_ids,
                                            d_out_dist, d_out_count, s);
}

extern "C" cuddle_status cuddle_sharded_search_batch(cuddle_index *shard, cuddle_comm *comm, const float *d_queries,
                                                     uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t probes_per_level,
                                                     uint32_t rerank_factor, uint32_t flags, uint64_t *d_out_ids,
                                                     float *d_out_dist, uint32_t *d_out_count, void *stream) {
    if (!shard || !comm || !d_queries || !d_out_ids || !d_out_dist) return fail(CUDDLE_ERR_BAD_ARG, "null argument");
    if (shard->device != comm->device) return fail(CUDDLE_ERR_BAD_ARG, "the shard lives on device %d, the communicator on %d", shard->device, comm->device);
    if (n_queries == 0) return fail(CUDDLE_ERR_BAD_ARG, "n_queries must be >= 1");
    if (dim != shard->dev.dim) return fail(CUDDLE_ERR_BAD_DIM, "query dimension %u != index dimension %u", dim, shard->dev.dim);
    if (k == 0 || (uint64_t)comm->world * k > 1024)
        return fail(CUDDLE_ERR_UNSUPPORTED, "the sharded search merges world * k <= 1024 results per query");
    std::lock_guard<std::mutex> lk(comm->mu);
    ST_TRY(use_device(comm->device));
    // larger batches run as equal rounds of at most kMaxChunk queries, one after the other on the stream (every rank
    // cuts the batch the same way: the cut depends on n_queries alone)
    const uint32_t rounds = (n_queries + kMaxChunk - 1) / kMaxChunk, per_round = (n_queries + rounds - 1) / rounds;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += per_round) {
        const uint32_t qc = std::min(per_round, n_queries - q0);
        ST_TRY(sharded_search_chunk(shard, comm, d_queries + (uint64_t)q0 * dim, qc, dim, k, probes_per_level, rerank_factor, flags,
                                    d_out_ids + (uint64_t)q0 * k, d_out_dist + (uint64_t)q0 * k, d_out_count ? d_out_count + q0 : nullptr,
                                    stream));
    }
    return CUDDLE_OK;
}
