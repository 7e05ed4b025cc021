"""Committed golden vectors (tests/golden): the oracle must reproduce them on CPU, the HIP path
must reproduce them on the GPU — bit for bit, stage by stage."""
import numpy as np
import pytest

import golden_util


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_oracle_reproduces_golden(name):
    from oracle import oracle as orc

    fi, q, par, exp = golden_util.load(name)
    oi = orc.OracleIndex(fi)
    for i in range(q.shape[0]):
        r = oi.search(q[i], par["k"], par["probes"], par["rf"], par["flags"], trace=True)
        assert r["rc"] == 0
        for key in ("leaves", "cand_pos", "cand_key", "sel_pos", "sel_key"):
            assert np.array_equal(r[key], exp[key][i]), (name, i, key)
        assert np.array_equal(_bits(r["sel_dist"]), _bits(exp["sel_dist"][i]))
        assert r["count"] == exp["count"][i]
        assert np.array_equal(r["ids"], exp["ids"][i]) and np.array_equal(_bits(r["dist"]), _bits(exp["dist"][i]))


def test_golden_set_is_present():
    assert len(golden_util.NAMES) >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_util.NAMES)
def test_gpu_reproduces_golden(gpu, name):
    from vectordb_amd.index import DeviceIndex

    fi, q, par, exp = golden_util.load(name)
    dev = DeviceIndex(fi, gpu)
    k, probes, rf, flags = par["k"], par["probes"], par["rf"], par["flags"]
    ids, dist, cnt = dev.search_batch(q, k, probes, rf, flags)
    assert np.array_equal(cnt, exp["count"])
    assert np.array_equal(ids, exp["ids"]) and np.array_equal(_bits(dist), _bits(exp["dist"]))
    leaves, lcnt = dev.route_batch(q, probes, flags)
    pos, key, ccnt = dev.scan_batch(q, leaves, lcnt)
    _, _, _, spos, skey, sdist, scnt = dev.search_leaves_batch(q, leaves, lcnt, k, rf, flags)
    for i in range(q.shape[0]):
        assert np.array_equal(leaves[i, :lcnt[i]], exp["leaves"][i]), (name, i)
        assert np.array_equal(pos[i, :ccnt[i]], exp["cand_pos"][i]) and np.array_equal(key[i, :ccnt[i]], exp["cand_key"][i])
        assert np.array_equal(spos[i, :scnt[i]], exp["sel_pos"][i]) and np.array_equal(skey[i, :scnt[i]], exp["sel_key"][i])
        assert np.array_equal(_bits(sdist[i, :scnt[i]]), _bits(exp["sel_dist"][i]))
