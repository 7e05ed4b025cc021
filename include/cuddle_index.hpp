// cuddle_index.hpp — C++ host side above the C ABI of libcuddle_hip.so (include/cuddle_hip.h).
//
// Mirrors the search surface of the reference's ClusteredIndex / ClusteredIndexWithRaBitQ
// (src/index/hierarchical.rs:330-336 search, :831-837 search_with_stats, :694-726 accessors;
// src/index/hierarchical_with_rabitq.rs:329-335): same method names, argument meaning, result
// shape (ascending (id, distance) pairs, possibly fewer than k) and error behaviour (the reference
// asserts on a dimension mismatch; here std::invalid_argument with the same message). The index
// itself arrives flattened (FlatIndex: the arrays of struct cuddle_index_desc), either built in
// memory or read from the flat-index file written by vectordb_amd.FlatIndex.save_bin — the
// reference has no index serialisation of its own (hierarchical.rs:466-481).
//
// Header-only, C++17; link with -lcuddle_hip. Fails loudly when the library or a GPU is missing:
// there is no CPU fallback.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "cuddle_hip.h"

namespace cuddle {

enum class DistanceMetric : uint32_t { L2 = CUDDLE_METRIC_L2, Cosine = CUDDLE_METRIC_COSINE, DotProduct = CUDDLE_METRIC_DOT }; // src/distance/mod.rs:10-17

// SearchStats (hierarchical.rs:36-45), field for field: leaves_searched_ids in accumulation order (:917),
// probes_per_level = the frontier size at every level including the empty frontier that ends the loop (:861),
// tree_depth = depth of the deepest node (the reference's max_depth, :851).
struct SearchStats {
    size_t total_leaves = 0;
    size_t leaves_searched = 0;
    std::vector<size_t> leaves_searched_ids;
    size_t total_vectors = 0;
    size_t vectors_scanned_binary = 0;
    size_t vectors_reranked_full = 0;
    size_t tree_depth = 0;
    std::vector<size_t> probes_per_level;
};

// The arrays of cuddle_index_desc, owned.
struct FlatIndex {
    uint32_t dim = 0, metric = 0, code_kind = 0;
    float threshold = 0.f;
    std::vector<float> centroids;
    std::vector<uint32_t> child_offsets, children, roots;
    std::vector<uint64_t> vec_offsets, ids;
    std::vector<float> vectors;
    std::vector<uint64_t> codes;
    std::vector<float> rabitq_rot_t;
    std::vector<uint32_t> rabitq_codes;
    std::vector<float> rabitq_norm, rabitq_ip_oo;

    uint32_t n_nodes() const { return child_offsets.empty() ? 0 : (uint32_t)child_offsets.size() - 1; }
    uint64_t n_vectors() const { return ids.size(); }

    cuddle_index_desc desc() const {
        cuddle_index_desc d;
        std::memset(&d, 0, sizeof(d));
        d.abi_version = CUDDLE_ABI_VERSION;
        d.dim = dim; d.metric = metric; d.code_kind = code_kind;
        d.n_nodes = n_nodes(); d.n_roots = (uint32_t)roots.size(); d.n_vectors = n_vectors();
        d.centroids = centroids.data(); d.child_offsets = child_offsets.data(); d.children = children.data();
        d.roots = roots.data(); d.vec_offsets = vec_offsets.data(); d.ids = ids.data(); d.vectors = vectors.data();
        d.codes = codes.empty() ? nullptr : codes.data();
        d.threshold = threshold;
        d.rabitq_rot_t = rabitq_rot_t.empty() ? nullptr : rabitq_rot_t.data();
        d.rabitq_codes = rabitq_codes.empty() ? nullptr : rabitq_codes.data();
        d.rabitq_norm = rabitq_norm.empty() ? nullptr : rabitq_norm.data();
        d.rabitq_ip_oo = rabitq_ip_oo.empty() ? nullptr : rabitq_ip_oo.data();
        return d;
    }

    // Flat-index file: "CUDDLEFI", u32 version = 1, u32 dim, metric, code_kind, n_nodes, n_roots,
    // n_children, u32 flags (bit 0 binary codes, bit 1 RaBitQ arrays), f32 threshold, u64 n_vectors,
    // then the arrays in the order of the struct, each padded to 8 bytes. Little-endian.
    static FlatIndex load(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        struct Closer { FILE *f; ~Closer() { std::fclose(f); } } closer{f};
        char magic[8];
        uint32_t h[8];
        float thr;
        uint64_t nvec;
        auto rd = [&](void *p, size_t bytes) {
            if (bytes && std::fread(p, 1, bytes, f) != bytes) throw std::runtime_error("short read in " + path);
        };
        rd(magic, 8);
        if (std::memcmp(magic, "CUDDLEFI", 8) != 0) throw std::runtime_error(path + " is not a flat-index file");
        rd(h, sizeof(h)); rd(&thr, 4);
        uint32_t pad4; rd(&pad4, 4);
        rd(&nvec, 8);
        if (h[0] != 1) throw std::runtime_error("unsupported flat-index version");
        FlatIndex x;
        x.dim = h[1]; x.metric = h[2]; x.code_kind = h[3]; x.threshold = thr;
        const uint32_t n_nodes = h[4], n_roots = h[5], n_children = h[6], flags = h[7];
        auto arr = [&](auto &v, size_t count) {
            v.resize(count);
            const size_t bytes = count * sizeof(v[0]);
            rd(v.data(), bytes);
            char skip[8];
            rd(skip, (8 - bytes % 8) % 8);
        };
        const size_t D = x.dim;
        arr(x.centroids, (size_t)n_nodes * D);
        arr(x.child_offsets, (size_t)n_nodes + 1);
        arr(x.children, n_children);
        arr(x.roots, n_roots);
        arr(x.vec_offsets, (size_t)n_nodes + 1);
        arr(x.ids, nvec);
        arr(x.vectors, nvec * D);
        if (flags & 1u) arr(x.codes, nvec * ((D + 63) / 64));
        if (flags & 2u) {
            arr(x.rabitq_rot_t, D * D);
            arr(x.rabitq_codes, nvec * ((D + 31) / 32));
            arr(x.rabitq_norm, nvec);
            arr(x.rabitq_ip_oo, nvec);
        }
        return x;
    }
};

class ClusteredIndex {
public:
    using Result = std::vector<std::pair<size_t, float>>;

    // Uploads the index to GPU `device` (the counterpart of build()/open() having run).
    explicit ClusteredIndex(const FlatIndex &flat, int device = 0)
        : dimension_(flat.dim), len_(flat.n_vectors()), nodes_(flat.n_nodes()) {
        for (uint32_t i = 0; i < flat.n_nodes(); ++i) leaves_ += flat.child_offsets[i + 1] == flat.child_offsets[i];
        const cuddle_index_desc d = flat.desc();
        check(cuddle_index_create(&d, device, &handle_));
    }
    ~ClusteredIndex() { if (handle_) cuddle_index_destroy(handle_); }
    ClusteredIndex(const ClusteredIndex &) = delete;
    ClusteredIndex &operator=(const ClusteredIndex &) = delete;
    ClusteredIndex(ClusteredIndex &&o) noexcept { *this = std::move(o); }
    ClusteredIndex &operator=(ClusteredIndex &&o) noexcept {
        std::swap(handle_, o.handle_); dimension_ = o.dimension_; len_ = o.len_; nodes_ = o.nodes_; leaves_ = o.leaves_;
        return *this;
    }

    size_t len() const { return len_; }                      // hierarchical.rs:694
    bool is_empty() const { return len_ == 0; }              // :699
    size_t dimension() const { return dimension_; }          // :704
    size_t max_depth() const { return cuddle_index_levels(handle_) - 1; } // :709
    size_t num_nodes() const { return nodes_; }              // :714
    size_t max_leaf_size() const { return cuddle_index_max_leaf_size(handle_); } // :719
    size_t count_leaves() const { return leaves_; }          // :724
    size_t memory_usage_bytes() const { return cuddle_index_device_bytes(handle_); } // :484 (device memory here)

    // search (hierarchical.rs:330-431): ascending (id, distance), at most k entries.
    Result search(const std::vector<float> &query, size_t k, size_t probes_per_level, size_t rerank_factor) const {
        return std::move(search_batch(query.data(), 1, query.size(), k, probes_per_level, rerank_factor, 0)[0]);
    }

    // search_with_stats (hierarchical.rs:831-938): the always-sorting variant plus its counters.
    std::pair<Result, SearchStats> search_with_stats(const std::vector<float> &query, size_t k, size_t probes_per_level,
                                                     size_t rerank_factor) const {
        if (query.size() != dimension_) throw std::invalid_argument("Query dimension mismatch"); // hierarchical.rs:838
        uint32_t lstride = 0;
        check(cuddle_index_leaf_stride(handle_, (uint32_t)probes_per_level, &lstride));
        const uint32_t levels = cuddle_index_levels(handle_);
        std::vector<uint64_t> ids(k);
        std::vector<float> dist(k);
        std::vector<uint32_t> leaves(lstride), level_nodes(levels + 1);
        uint32_t count = 0, n_leaves = 0, n_levels = 0, scanned = 0, reranked = 0;
        check(cuddle_search_batch_with_stats(handle_, query.data(), 1, (uint32_t)query.size(), (uint32_t)k,
                                             (uint32_t)probes_per_level, (uint32_t)rerank_factor, ids.data(), dist.data(), &count,
                                             leaves.data(), lstride, &n_leaves, level_nodes.data(), levels + 1, &n_levels,
                                             &scanned, &reranked));
        Result r;
        for (uint32_t i = 0; i < count; ++i) r.emplace_back((size_t)ids[i], dist[i]);
        SearchStats out;
        out.total_leaves = leaves_;
        out.total_vectors = len_;
        out.leaves_searched = n_leaves;
        out.leaves_searched_ids.assign(leaves.begin(), leaves.begin() + n_leaves);
        out.vectors_scanned_binary = scanned;
        out.vectors_reranked_full = reranked;
        out.tree_depth = max_depth();
        out.probes_per_level.assign(level_nodes.begin(), level_nodes.begin() + n_levels);
        return {std::move(r), std::move(out)};
    }

    // One search() per row of `queries` ([n_queries][dim] row-major) in a single batch on the GPU.
    std::vector<Result> search_batch(const float *queries, size_t n_queries, size_t dim, size_t k, size_t probes_per_level,
                                     size_t rerank_factor, uint32_t flags = 0, cuddle_stats *stats = nullptr) const {
        if (dim != dimension_) throw std::invalid_argument("Query dimension mismatch"); // hierarchical.rs:337
        std::vector<uint64_t> ids(n_queries * k);
        std::vector<float> dist(n_queries * k);
        std::vector<uint32_t> count(n_queries);
        check(cuddle_search_batch(handle_, queries, (uint32_t)n_queries, (uint32_t)dim, (uint32_t)k, (uint32_t)probes_per_level,
                                  (uint32_t)rerank_factor, flags, ids.data(), dist.data(), count.data(), stats));
        std::vector<Result> out(n_queries);
        for (size_t q = 0; q < n_queries; ++q) {
            out[q].reserve(count[q]);
            for (uint32_t i = 0; i < count[q]; ++i) out[q].emplace_back((size_t)ids[q * k + i], dist[q * k + i]);
        }
        return out;
    }

    cuddle_index *handle() const { return handle_; }

private:
    static void check(cuddle_status st) {
        if (st == CUDDLE_OK) return;
        const std::string msg = std::string("libcuddle_hip: ") + cuddle_last_error();
        if (st == CUDDLE_ERR_BAD_ARG || st == CUDDLE_ERR_BAD_DIM) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }
    cuddle_index *handle_ = nullptr;
    size_t dimension_ = 0, len_ = 0, nodes_ = 0, leaves_ = 0;
};

// The RaBitQ twin shares the surface (hierarchical_with_rabitq.rs:329-335): the code kind is a
// property of the flat index.
using ClusteredIndexWithRaBitQ = ClusteredIndex;

} // namespace cuddle
