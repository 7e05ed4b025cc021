"""The C++ host side (include/cuddle_index.hpp): flat-index file round trip and header syntax on CPU;
on the GPU the compiled test program must reproduce the oracle's results through the C++ class."""
import os
import subprocess

import numpy as np
import pytest

from vectordb_amd.flat_index import FlatIndex

from helpers import queries_like, small_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def build_cpp():
    """g++ only: the program sees libcuddle_hip.so through its C header."""
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), "-o", BIN,
           "-L", os.path.join(ROOT, "vectordb_amd"), "-lcuddle_hip", "-Wl,-rpath," + os.path.join(ROOT, "vectordb_amd")]
    subprocess.run(cmd, check=True)


@pytest.mark.parametrize("code_kind", ["binary", "rabitq"])
def test_flat_index_file_round_trip(tmp_path, code_kind):
    x, fi = small_index(n=1500, d=24, n_clusters=10, branching=5, leaf=25, code_kind=code_kind)
    p = str(tmp_path / "i.cfi")
    fi.save_bin(p)
    g = FlatIndex.load_bin(p)
    for name in ("centroids", "child_offsets", "children", "roots", "vec_offsets", "ids", "vectors", "codes",
                 "rabitq_rot_t", "rabitq_codes", "rabitq_norm", "rabitq_ip_oo"):
        a, b = getattr(fi, name), getattr(g, name)
        assert (a is None) == (b is None), name
        if a is not None:
            assert np.array_equal(a, b), name
    assert (g.dim, g.metric, g.code_kind) == (fi.dim, fi.metric, fi.code_kind) and g.threshold == np.float32(fi.threshold)


def test_cpp_host_compiles_against_the_library():
    build_cpp()
    assert os.path.exists(BIN)


def _write_cases(path, q, k, probes, rf, stats, oi):
    nq, dim = q.shape
    cnt = np.zeros(nq, dtype=np.uint32)
    ids = np.full((nq, k), np.iinfo(np.uint64).max, dtype=np.uint64)
    dd = np.full((nq, k), np.inf, dtype=np.float32)
    for i in range(nq):
        r = oi.search(q[i], k, probes, rf, 1 if stats else 0)
        assert r["rc"] == 0
        cnt[i] = r["count"]
        ids[i, :cnt[i]], dd[i, :cnt[i]] = r["ids"][:cnt[i]], r["dist"][:cnt[i]]
    with open(path, "wb") as f:
        f.write(np.array([nq, dim, k, probes, rf, 1 if stats else 0], dtype="<u4").tobytes())
        f.write(q.astype("<f4").tobytes() + cnt.tobytes() + ids.tobytes() + dd.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("metric,code_kind,stats", [("l2", "binary", 0), ("cosine", "binary", 1), ("dot", "rabitq", 0)])
def test_cpp_host_matches_oracle(gpu, tmp_path, metric, code_kind, stats):
    from oracle import oracle as orc
    if not os.path.exists(BIN):
        build_cpp()
    x, fi = small_index(n=3000, d=32, n_clusters=20, branching=6, leaf=30, metric=metric, code_kind=code_kind)
    q = queries_like(x, 20, 5)
    ip, cp = str(tmp_path / "i.cfi"), str(tmp_path / "c.bin")
    fi.save_bin(ip)
    _write_cases(cp, q, 10, 4, 5, stats, orc.OracleIndex(fi))
    r = subprocess.run([BIN, ip, cp], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
