"""Per-phase timing of the grouped routing kernel's blocks (needs a -DCUDDLE_GROUP_TIMING build of
the library, selected with CUDDLE_HIP_LIB). Development tool; not used by tests or the benchmark."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vectordb_amd import _ffi  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "shard12m"
    cfg = bench.CONFIGS[name]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from vectordb_amd.synthetic import build_tree_index_on_device
    fanout, levels, leaf = cfg["tree"]
    dti, queries = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, device, seed=0xC0DD1E,
                                              n_queries=cfg["n_queries"])
    idx = dti.create(0)
    nq, k = queries.shape[0], cfg["k"]
    ids = torch.empty((nq, k), dtype=torch.int64, device=device)
    dd = torch.empty((nq, k), dtype=torch.float32, device=device)
    cnt = torch.empty((nq,), dtype=torch.int32, device=device)
    for _ in range(3):
        idx.search_batch_device(queries, k, cfg["probes"], 25, 0, ids, dd, cnt)
    torch.cuda.synchronize()
    fn = _ffi.lib().cuddle_debug_group_times
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    fn.restype = ctypes.c_int
    out = np.zeros((16384, 4), dtype=np.uint64)
    assert fn(out.ctypes.data, 16384) == 0
    t = out.astype(np.int64)
    t = t[t[:, 3] > 0]
    d = np.diff(t, axis=1) * 0.01
    for i, nm in enumerate(["prologue + rows", "first tile staged", "tiles"]):
        print(f"{nm:20s} mean {d[:, i].mean():8.2f} us  p50 {np.median(d[:, i]):8.2f}  max {d[:, i].max():8.2f}")
    life = (t[:, 3] - t[:, 0]) * 0.01
    span = (t[:, 3].max() - t[:, 0].min()) * 0.01
    print(f"blocks {len(t)}; lifetime mean {life.mean():.2f} us; span {span:.1f} us; in flight (avg) {life.sum() / span:.1f}")


if __name__ == "__main__":
    main()
