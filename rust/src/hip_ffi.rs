//! `extern "C"` view of include/cuddle_hip.h (ABI version 2). Field order and types follow the header.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const CUDDLE_ABI_VERSION: u32 = 2;
pub const CUDDLE_OK: c_int = 0;
pub const CUDDLE_SEARCH_STATS_VARIANT: u32 = 0x1;
pub const CUDDLE_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct CuddleIndexDesc {
    pub abi_version: u32,
    pub dim: u32,
    pub metric: u32,    // DistanceMetric: L2 = 0, Cosine = 1, DotProduct = 2
    pub code_kind: u32, // 0 binary codes, 1 RaBitQ codes
    pub n_nodes: u32,
    pub n_roots: u32,
    pub n_vectors: u64,
    pub centroids: *const f32,
    pub child_offsets: *const u32,
    pub children: *const u32,
    pub roots: *const u32,
    pub vec_offsets: *const u64,
    pub ids: *const u64,
    pub vectors: *const f32,
    pub codes: *const u64,
    pub threshold: f32,
    pub flags: u32,
    pub rabitq_rot_t: *const f32,
    pub rabitq_codes: *const u32,
    pub rabitq_norm: *const f32,
    pub rabitq_ip_oo: *const f32,
}

#[repr(C)]
pub struct CuddleIndex {
    _private: [u8; 0],
}
#[repr(C)]
pub struct CuddleComm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct CuddleStats {
    pub leaves_searched: u64,
    pub vectors_scanned: u64,
    pub vectors_reranked: u64,
    pub max_candidates: u32,
    pub max_leaves: u32,
    pub tree_levels: u32,
    pub route_fallbacks: u32,
    pub rabitq_exact_estimates: u64,
    pub rabitq_exact_all_queries: u64,
}

extern "C" {
    pub fn cuddle_abi_version() -> u32;
    pub fn cuddle_last_error() -> *const c_char;
    pub fn cuddle_device_count() -> c_int;
    pub fn cuddle_index_create(desc: *const CuddleIndexDesc, device_id: c_int, out: *mut *mut CuddleIndex) -> c_int;
    pub fn cuddle_index_destroy(idx: *mut CuddleIndex);
    pub fn cuddle_index_reserve(idx: *mut CuddleIndex, max_queries: u32, max_k: u32, max_probes: u32, max_rerank_k: u32) -> c_int;
    pub fn cuddle_index_levels(idx: *const CuddleIndex) -> u32;
    pub fn cuddle_index_max_leaf_size(idx: *const CuddleIndex) -> u32;
    pub fn cuddle_index_leaf_stride(idx: *const CuddleIndex, probes_per_level: u32, out: *mut u32) -> c_int;
    pub fn cuddle_search_batch(
        idx: *mut CuddleIndex, queries: *const f32, n_queries: u32, dim: u32, k: u32, probes_per_level: u32,
        rerank_factor: u32, flags: u32, out_ids: *mut u64, out_dist: *mut f32, out_count: *mut u32,
        stats: *mut CuddleStats,
    ) -> c_int;
    pub fn cuddle_search_batch_device(
        idx: *mut CuddleIndex, d_queries: *const f32, n_queries: u32, dim: u32, k: u32, probes_per_level: u32,
        rerank_factor: u32, flags: u32, d_out_ids: *mut u64, d_out_dist: *mut f32, d_out_count: *mut u32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn cuddle_search_batch_device_noalloc(
        idx: *mut CuddleIndex, d_queries: *const f32, n_queries: u32, dim: u32, k: u32, probes_per_level: u32,
        rerank_factor: u32, flags: u32, d_out_ids: *mut u64, d_out_dist: *mut f32, d_out_count: *mut u32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn cuddle_search_batch_with_stats(
        idx: *mut CuddleIndex, queries: *const f32, n_queries: u32, dim: u32, k: u32, probes: u32, rerank_factor: u32,
        out_ids: *mut u64, out_dist: *mut f32, out_count: *mut u32, out_leaves: *mut u32, leaf_stride: u32,
        out_leaf_count: *mut u32, out_level_nodes: *mut u32, level_stride: u32, out_level_count: *mut u32,
        out_scanned: *mut u32, out_reranked: *mut u32,
    ) -> c_int;
    // index build: KMeans::assign (src/clustering/kmeans.rs:136-160) on device-resident data
    pub fn cuddle_kmeans_assign(
        d_vectors: *const f32, n: u64, d_centroids: *const f32, k: u32, dim: u32, metric: u32, d_assign: *mut u32,
        d_dist: *mut f32, out_fallbacks: *mut u64, device_id: c_int, stream: *mut c_void,
    ) -> c_int;
    // one process per GPU: RCCL communicator + the exact sharded search
    pub fn cuddle_comm_unique_id(id: *mut u8) -> c_int;
    pub fn cuddle_comm_init(rank: c_int, world: c_int, id: *const u8, device_id: c_int, out: *mut *mut CuddleComm) -> c_int;
    pub fn cuddle_comm_destroy(comm: *mut CuddleComm);
    pub fn cuddle_sharded_search_batch(
        shard: *mut CuddleIndex, comm: *mut CuddleComm, d_queries: *const f32, n_queries: u32, dim: u32, k: u32,
        probes_per_level: u32, rerank_factor: u32, flags: u32, d_out_ids: *mut u64, d_out_dist: *mut f32,
        d_out_count: *mut u32, stream: *mut c_void,
    ) -> c_int;
}

pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(cuddle_last_error()).to_string_lossy().into_owned() }
}
