// Exercises the C++ host side (include/cuddle_index.hpp) the way the reference's own tests use
// ClusteredIndex: load an index, search, compare with expected (id, distance) lists — here the
// oracle's, written by tests/test_cpp_host.py. Usage:
//   host_mirror_test <index.cfi> <cases.bin>
// cases.bin: u32 n_queries, dim, k, probes, rerank_factor, stats_variant; f32 queries[nq][dim];
//            u32 count[nq]; u64 ids[nq][k]; f32 dist[nq][k]  (expected)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuddle_index.hpp"

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s index.cfi cases.bin\n", argv[0]); return 2; }
    try {
        cuddle::FlatIndex flat = cuddle::FlatIndex::load(argv[1]);
        FILE *f = std::fopen(argv[2], "rb");
        if (!f) { std::perror(argv[2]); return 2; }
        uint32_t h[6];
        if (std::fread(h, 4, 6, f) != 6) return 2;
        const uint32_t nq = h[0], dim = h[1], k = h[2], probes = h[3], rf = h[4], stats = h[5];
        std::vector<float> q((size_t)nq * dim), ed((size_t)nq * k);
        std::vector<uint32_t> ec(nq);
        std::vector<uint64_t> ei((size_t)nq * k);
        if (std::fread(q.data(), 4, q.size(), f) != q.size() || std::fread(ec.data(), 4, nq, f) != nq ||
            std::fread(ei.data(), 8, ei.size(), f) != ei.size() || std::fread(ed.data(), 4, ed.size(), f) != ed.size()) return 2;
        std::fclose(f);

        cuddle::ClusteredIndex index(flat, 0);
        if (index.len() != flat.n_vectors() || index.dimension() != dim) { std::fprintf(stderr, "accessors\n"); return 1; }
        int bad = 0;
        // one query at a time, like the reference's callers ...
        for (uint32_t i = 0; i < nq && !bad; ++i) {
            std::vector<float> one(q.begin() + (size_t)i * dim, q.begin() + (size_t)(i + 1) * dim);
            cuddle::ClusteredIndex::Result r;
            if (stats) {
                auto rs = index.search_with_stats(one, k, probes, rf);
                r = rs.first;
                const cuddle::SearchStats &st = rs.second;
                if (st.leaves_searched != st.leaves_searched_ids.size() || st.total_vectors != index.len()) bad = 1;
                // hierarchical.rs:861: one entry per loop iteration, the roots first, the empty frontier last
                if (st.probes_per_level.empty() || st.probes_per_level.front() != flat.roots.size() || st.probes_per_level.back() != 0) bad = 1;
                if (st.tree_depth != index.max_depth() || st.total_leaves != index.count_leaves()) bad = 1;
                if (st.vectors_reranked_full != std::min<size_t>((size_t)k * rf, st.vectors_scanned_binary)) bad = 1;
            } else {
                r = index.search(one, k, probes, rf);
            }
            if (r.size() != ec[i]) bad = 1;
            for (size_t j = 0; j < r.size() && !bad; ++j)
                if (r[j].first != ei[(size_t)i * k + j] || std::memcmp(&r[j].second, &ed[(size_t)i * k + j], 4) != 0) bad = 1;
            if (bad) std::fprintf(stderr, "mismatch at query %u\n", i);
        }
        // ... and as one batch
        if (!bad && !stats) {
            auto all = index.search_batch(q.data(), nq, dim, k, probes, rf);
            for (uint32_t i = 0; i < nq && !bad; ++i) {
                if (all[i].size() != ec[i]) bad = 1;
                for (size_t j = 0; j < all[i].size() && !bad; ++j)
                    if (all[i][j].first != ei[(size_t)i * k + j] || std::memcmp(&all[i][j].second, &ed[(size_t)i * k + j], 4) != 0) bad = 1;
            }
            if (bad) std::fprintf(stderr, "batch mismatch\n");
        }
        // error behaviour: wrong dimension -> invalid_argument("Query dimension mismatch")
        if (!bad) {
            bool threw = false;
            try { index.search(std::vector<float>(dim + 1, 0.f), k, probes, rf); } catch (const std::invalid_argument &) { threw = true; }
            if (!threw) { std::fprintf(stderr, "no exception on dimension mismatch\n"); bad = 1; }
        }
        std::printf(bad ? "FAIL\n" : "OK %u queries\n", nq);
        return bad;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
}
