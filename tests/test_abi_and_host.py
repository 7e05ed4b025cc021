"""CPU-side checks: the C-ABI library loads and exports every symbol include/cuddle_hip.h
declares, refuses to compute without a GPU (no fallback), and the host-side helpers (code
packing, threshold, leaf sharding, top-k merge, index builder) are consistent with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vectordb_amd import _ffi
from vectordb_amd.builder import build_index
from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
from oracle import oracle as orc

from helpers import clustered, queries_like, small_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "cuddle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cuddle_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 20
    h = C.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"libcuddle_hip.so does not export {n}"
    assert sorted(_ffi.SIGNATURES) == names, "python binding and header disagree"
    assert _ffi.lib().cuddle_abi_version() == _ffi.ABI_VERSION


def test_struct_layout_matches_header():
    assert C.sizeof(_ffi.IndexDesc) == 32 + 8 * 8 + 8 + 4 * 8  # header fields, 8-byte aligned
    assert C.sizeof(_ffi.Stats) == 56


def test_no_gpu_fails_loudly():
    if _ffi.lib().cuddle_device_count() > 0:
        pytest.skip("a GPU is visible here")
    x, fi = small_index(n=400, d=8, n_clusters=4, branching=4, leaf=20)
    from vectordb_amd.index import DeviceIndex
    with pytest.raises(_ffi.CuddleError) as e:
        DeviceIndex(fi, 0)
    assert e.value.status == _ffi.ERR_NO_DEVICE
    out = np.zeros(1, dtype=np.float32)
    a = np.ones((1, 4), dtype=np.float32)
    st = _ffi.lib().cuddle_distance_pairs(_ffi.ptr(a), _ffi.ptr(a), 1, 4, 0, _ffi.ptr(out), 0)
    assert st == _ffi.ERR_NO_DEVICE


def test_bad_desc_rejected_before_touching_the_gpu():
    x, fi = small_index(n=400, d=8, n_clusters=4, branching=4, leaf=20)
    fi.vec_offsets = fi.vec_offsets.copy()
    fi.vec_offsets[-1] += 1
    desc = fi.desc()
    h = C.c_void_p(None)
    st = _ffi.lib().cuddle_index_create(C.byref(desc), 0, C.byref(h))
    assert st == _ffi.ERR_BAD_INDEX and not h.value


def test_pack_codes_and_threshold_match_oracle():
    rng = np.random.default_rng(1)
    for d in (1, 63, 64, 65, 128, 960):
        v = rng.standard_normal((37, d)).astype(np.float32)
        assert np.array_equal(pack_binary_codes(v, 0.1), orc.binary_quantize(v, 0.1))
    v = (rng.random((5000, 128)) * 255).astype(np.float32)
    lib_thr = float(_ffi.lib().cuddle_binary_threshold(_ffi.ptr(v), v.shape[0], v.shape[1]))
    assert lib_thr == orc.binary_threshold(v)


def test_builder_invariants():
    x, fi = small_index(n=6000, d=16, n_clusters=30, branching=8, leaf=40)
    assert sorted(fi.ids.tolist()) == list(range(6000))
    assert np.array_equal(fi.vectors, x[fi.ids.astype(np.int64)])
    assert fi.leaf_sizes().sum() == 6000
    leaf = fi.is_leaf()
    sizes = (fi.vec_offsets[1:] - fi.vec_offsets[:-1]).astype(np.int64)
    assert np.all(sizes[~leaf] == 0)
    # every non-root node has exactly one parent
    assert sorted(fi.children.tolist() + fi.roots.tolist()) == list(range(fi.n_nodes))
    fi2 = build_index(x, 8, 40, "l2", 10, seed=1)
    assert np.array_equal(fi.ids, fi2.ids) and np.array_equal(fi.centroids, fi2.centroids)  # deterministic


def test_save_load_roundtrip(tmp_path):
    x, fi = small_index(n=500, d=8, n_clusters=5, branching=4, leaf=20)
    p = str(tmp_path / "idx.npz")
    fi.save(p)
    g = FlatIndex.load(p)
    for k in ("centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes"):
        assert np.array_equal(getattr(fi, k), getattr(g, k)), k
    assert g.threshold == fi.threshold and g.dim == fi.dim


def test_merge_topk_host():
    inf, m = np.float32(np.inf), np.uint64(0xFFFFFFFFFFFFFFFF)
    ids = np.array([[[1, 2, m]], [[3, 4, 5]]], dtype=np.uint64)          # [lists=2][q=1][k=3]
    dist = np.array([[[0.5, 2.0, inf]], [[0.5, 1.0, 3.0]]], dtype=np.float32)
    oi = np.zeros((1, 3), dtype=np.uint64)
    od = np.zeros((1, 3), dtype=np.float32)
    oc = np.zeros(1, dtype=np.uint32)
    _ffi.check(_ffi.lib().cuddle_merge_topk(_ffi.ptr(ids), _ffi.ptr(dist), 2, 1, 3, _ffi.ptr(oi), _ffi.ptr(od), _ffi.ptr(oc)))
    assert oi.tolist() == [[1, 3, 4]] and od.tolist() == [[0.5, 0.5, 1.0]] and oc[0] == 3


def test_leaf_shards_partition_the_index_and_merge_back():
    """Leaf sharding: the shards' vectors partition the index; with rerank_k >= every candidate
    the merged per-shard oracle answers equal the unsharded oracle answer bit for bit."""
    x, fi = small_index(n=4000, d=16, n_clusters=20, branching=6, leaf=30)
    world = 3
    shards = [fi.shard(r, world) for r in range(world)]
    assert sum(s.n_vectors for s in shards) == fi.n_vectors
    assert sorted(np.concatenate([s.ids for s in shards]).tolist()) == list(range(4000))
    q = queries_like(x, 16, 9)
    k, probes, rf = 10, 4, 1000
    full = orc.OracleIndex(fi)
    parts = [orc.OracleIndex(s) for s in shards]
    for i in range(q.shape[0]):
        ref = full.search(q[i], k, probes, rf)
        got = [p.search(q[i], k, probes, rf) for p in parts]
        ids = np.stack([g["ids"] for g in got])[:, None, :].copy()
        dist = np.stack([g["dist"] for g in got])[:, None, :].copy()
        oi = np.zeros((1, k), dtype=np.uint64)
        od = np.zeros((1, k), dtype=np.float32)
        _ffi.check(_ffi.lib().cuddle_merge_topk(_ffi.ptr(ids), _ffi.ptr(dist), world, 1, k, _ffi.ptr(oi), _ffi.ptr(od), None))
        assert np.array_equal(od[0], ref["dist"])
        assert sorted(oi[0].tolist()) == sorted(ref["ids"].tolist())


def test_two_hip_runtimes_are_named_not_left_to_fail_later():
    """Loading the library before torch leaves two libamdhip64 copies mapped (torch loads its own by file name);
    the library then refuses to create an index and names both files, instead of torch reporting "no ROCm-capable
    device" much later. Runs in a child process: the load order of this one is already decided."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
lib = ctypes.CDLL(%r)            # the library first: bound to the system runtime
import torch                      # torch maps its own copy
import numpy as np
from vectordb_amd import _ffi
from vectordb_amd.flat_index import FlatIndex, pack_binary_codes
x = np.zeros((4, 8), np.float32)
fi = FlatIndex(dim=8, metric=0, code_kind=0, centroids=x[:1].copy(), child_offsets=np.zeros(2, np.uint32),
               children=np.zeros(0, np.uint32), roots=np.zeros(1, np.uint32), vec_offsets=np.array([0, 4], np.uint64),
               ids=np.arange(4, dtype=np.uint64), vectors=x, threshold=0.0, codes=pack_binary_codes(x, 0.0)).normalised()
h = ctypes.c_void_p(None)
d = fi.desc()
st = _ffi.lib().cuddle_index_create(ctypes.byref(d), 0, ctypes.byref(h))
print(st, _ffi.lib().cuddle_last_error().decode())
''' % (str(ROOT), str(_ffi.LIB_PATH))
    maps_has_two = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert maps_has_two.returncode == 0, maps_has_two.stderr[-2000:]
    status, _, msg = maps_has_two.stdout.strip().partition(" ")
    if "two HIP runtimes" not in msg:  # a torch build that uses the system runtime: nothing to detect
        pytest.skip("torch shares the system HIP runtime here: " + msg[:200])
    assert int(status) == _ffi.ERR_HIP
    assert msg.count("libamdhip64") >= 2 and "torch" in msg
