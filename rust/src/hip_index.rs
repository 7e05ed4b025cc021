//! The GPU-resident twin of `ClusteredIndex` / `ClusteredIndexWithRaBitQ`: same `search` contract
//! (src/index/hierarchical.rs:330-336), plus the batched form the reference only sketches
//! (benches/profile_bench.rs:286-296). Goes into src/index/ of the reference crate (feature `hip`).
use super::hierarchical::{ClusteredIndex, SearchStats};
use super::hip_ffi::*;

pub struct HipIndex {
    handle: *mut CuddleIndex,
    dim: usize,
    total_leaves: usize,
    total_vectors: usize,
}
unsafe impl Send for HipIndex {} // calls on one handle are serialised inside the library

impl ClusteredIndex {
    /// Upload once after build(); the handle lives next to the index. Arrays are flattened in
    /// leaf-contiguous order: leaves in node-id order, each leaf's `vector_indices` in stored order.
    pub fn to_hip(&self, device_id: i32) -> HipIndex {
        let d = self.dimension;
        let n = self.nodes.len();
        let mut centroids = Vec::with_capacity(n * d);
        let (mut child_off, mut children) = (vec![0u32], Vec::new());
        let (mut vec_off, mut ids) = (vec![0u64], Vec::new());
        for node in &self.nodes {
            // node id == position in `nodes` (hierarchical.rs:211-216, 304-316)
            centroids.extend_from_slice(&node.full_centroid);
            children.extend(node.children.iter().map(|&c| c as u32));
            child_off.push(children.len() as u32);
            ids.extend(node.vector_indices.iter().map(|&i| i as u64));
            vec_off.push(ids.len() as u64);
        }
        let words = (d + 63) / 64;
        let mut vectors = Vec::with_capacity(ids.len() * d);
        let mut codes = Vec::with_capacity(ids.len() * words);
        for &i in &ids {
            vectors.extend_from_slice(self.full_vectors.get(i as usize)); // storage/mmap.rs:106-115
            codes.extend_from_slice(&self.binary_vectors[i as usize].bits); // quantization/binary.rs:16-21
        }
        let roots: Vec<u32> = self.root_ids.iter().map(|&r| r as u32).collect();
        let desc = CuddleIndexDesc {
            abi_version: CUDDLE_ABI_VERSION,
            dim: d as u32,
            metric: self.metric as u32,
            code_kind: 0,
            n_nodes: n as u32,
            n_roots: roots.len() as u32,
            n_vectors: ids.len() as u64,
            centroids: centroids.as_ptr(),
            child_offsets: child_off.as_ptr(),
            children: children.as_ptr(),
            roots: roots.as_ptr(),
            vec_offsets: vec_off.as_ptr(),
            ids: ids.as_ptr(),
            vectors: vectors.as_ptr(),
            codes: codes.as_ptr(),
            threshold: self.quantizer.threshold(),
            flags: 0,
            rabitq_rot_t: std::ptr::null(),
            rabitq_codes: std::ptr::null(),
            rabitq_norm: std::ptr::null(),
            rabitq_ip_oo: std::ptr::null(),
        };
        let mut handle = std::ptr::null_mut();
        let st = unsafe { cuddle_index_create(&desc, device_id, &mut handle) };
        assert_eq!(st, CUDDLE_OK, "{}", last_error());
        HipIndex { handle, dim: d, total_leaves: self.count_leaves(), total_vectors: ids.len() }
    }
}
// ClusteredIndexWithRaBitQ::to_hip is the same with code_kind = 1, codes = null, rabitq_rot_t = the rows of
// rotation_matrix_t flattened (quantization/rabitq.rs:20; needs a pub(crate) accessor) and
// rabitq_codes / rabitq_norm / rabitq_ip_oo from RaBitQVector (:24-32) in leaf-contiguous order.

impl HipIndex {
    /// Same contract as ClusteredIndex::search: ascending (id, distance), at most k entries.
    pub fn search(&self, query: &[f32], k: usize, probes_per_level: usize, rerank_factor: usize) -> Vec<(usize, f32)> {
        self.search_batch(query, 1, k, probes_per_level, rerank_factor).pop().unwrap()
    }

    /// One search() per row of `queries` ([nq][dim] row-major) in a single batch on the GPU.
    pub fn search_batch(&self, queries: &[f32], nq: usize, k: usize, probes: usize, rf: usize) -> Vec<Vec<(usize, f32)>> {
        assert_eq!(queries.len(), nq * self.dim, "Query dimension mismatch"); // hierarchical.rs:337
        let (mut ids, mut dist, mut cnt) = (vec![u64::MAX; nq * k], vec![f32::INFINITY; nq * k], vec![0u32; nq]);
        let st = unsafe {
            cuddle_search_batch(self.handle, queries.as_ptr(), nq as u32, self.dim as u32, k as u32, probes as u32,
                                rf as u32, 0, ids.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr(), std::ptr::null_mut())
        };
        assert_eq!(st, CUDDLE_OK, "{}", last_error());
        (0..nq).map(|q| (0..cnt[q] as usize).map(|i| (ids[q * k + i] as usize, dist[q * k + i])).collect()).collect()
    }

    /// search_with_stats (hierarchical.rs:831-938), SearchStats field for field.
    pub fn search_with_stats(&self, query: &[f32], k: usize, probes: usize, rf: usize) -> (Vec<(usize, f32)>, SearchStats) {
        assert_eq!(query.len(), self.dim);
        let levels = unsafe { cuddle_index_levels(self.handle) } as usize;
        let mut lstride = 0u32;
        unsafe { cuddle_index_leaf_stride(self.handle, probes as u32, &mut lstride) };
        let (mut ids, mut dist, mut cnt) = (vec![u64::MAX; k], vec![f32::INFINITY; k], 0u32);
        let (mut leaves, mut n_leaves) = (vec![0u32; lstride as usize], 0u32);
        let (mut level_nodes, mut n_levels) = (vec![0u32; levels + 1], 0u32);
        let (mut scanned, mut reranked) = (0u32, 0u32);
        let st = unsafe {
            cuddle_search_batch_with_stats(self.handle, query.as_ptr(), 1, self.dim as u32, k as u32, probes as u32, rf as u32,
                                           ids.as_mut_ptr(), dist.as_mut_ptr(), &mut cnt, leaves.as_mut_ptr(), lstride,
                                           &mut n_leaves, level_nodes.as_mut_ptr(), (levels + 1) as u32, &mut n_levels,
                                           &mut scanned, &mut reranked)
        };
        assert_eq!(st, CUDDLE_OK, "{}", last_error());
        let results = (0..cnt as usize).map(|i| (ids[i] as usize, dist[i])).collect();
        let stats = SearchStats {
            total_leaves: self.total_leaves,
            leaves_searched: n_leaves as usize,
            leaves_searched_ids: leaves[..n_leaves as usize].iter().map(|&l| l as usize).collect(),
            total_vectors: self.total_vectors,
            vectors_scanned_binary: scanned as usize,
            vectors_reranked_full: reranked as usize,
            tree_depth: levels.saturating_sub(1),
            probes_per_level: level_nodes[..n_levels as usize].iter().map(|&v| v as usize).collect(),
        };
        (results, stats)
    }
}

impl Drop for HipIndex {
    fn drop(&mut self) {
        unsafe { cuddle_index_destroy(self.handle) }
    }
}

/// One process per GPU: this rank's leaf shard + an RCCL communicator. `unique_id` is made by rank 0
/// (`cuddle_comm_unique_id`) and handed to the other ranks by the application.
pub struct ShardedHipIndex {
    shard: HipIndex,
    comm: *mut CuddleComm,
}

impl ShardedHipIndex {
    pub fn new(shard: HipIndex, rank: i32, world: i32, unique_id: &[u8; CUDDLE_COMM_ID_BYTES], device_id: i32) -> Self {
        let mut comm = std::ptr::null_mut();
        let st = unsafe { cuddle_comm_init(rank, world, unique_id.as_ptr(), device_id, &mut comm) };
        assert_eq!(st, CUDDLE_OK, "{}", last_error());
        ShardedHipIndex { shard, comm }
    }

    /// Every rank calls this with the same device-resident queries and receives the whole index's result.
    pub unsafe fn search_batch_device(&self, d_queries: *const f32, nq: u32, k: u32, probes: u32, rf: u32, d_ids: *mut u64,
                                      d_dist: *mut f32, d_count: *mut u32, stream: *mut std::ffi::c_void) {
        let st = cuddle_sharded_search_batch(self.shard.handle, self.comm, d_queries, nq, self.shard.dim as u32, k, probes, rf,
                                             0, d_ids, d_dist, d_count, stream);
        assert_eq!(st, CUDDLE_OK, "{}", last_error());
    }
}

impl Drop for ShardedHipIndex {
    fn drop(&mut self) {
        unsafe { cuddle_comm_destroy(self.comm) }
    }
}
