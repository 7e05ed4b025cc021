"""Per-phase timing of the blocks of one kernel. Needs a build of the library with
-DCUDDLE_STAMP_KERNEL=<id> (1 Hamming scan, 2 grouped routing distances, 3 register selection, 6 RaBitQ scan),
selected with CUDDLE_HIP_LIB; prints the mean time between consecutive stamps of a block, the block
lifetime and the average number of blocks in flight. Development tool; not used by the tests or
the benchmark.

    hipcc ... -DCUDDLE_STAMP_KERNEL=2 -o vectordb_amd/libcuddle_timing.so vectordb_amd/csrc/cuddle_hip.hip
    CUDDLE_HIP_LIB=$PWD/vectordb_amd/libcuddle_timing.so python tools/kernel_phase_times.py shard12m 25 4
"""
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
    rf = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    npts = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cfg = bench.CONFIGS[name]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if "tree" in cfg:
        from vectordb_amd.synthetic import build_tree_index_on_device
        fanout, levels, leaf = cfg["tree"]
        from vectordb_amd.builder import metric_id
        dti, queries = build_tree_index_on_device(cfg["dim"], fanout, levels, leaf, device, seed=0xC0DD1E,
                                                  n_queries=cfg["n_queries"], metric=metric_id(cfg["metric"]),
                                                  style=cfg.get("style", "sift"),
                                                  code_kind=_ffi.CODE_RABITQ if cfg["code_kind"] == "rabitq" else _ffi.CODE_BINARY)
        idx = dti.create(0)
    else:
        from vectordb_amd.builder import build_index
        from vectordb_amd.datasets import synthetic_mixture
        from vectordb_amd.index import DeviceIndex
        base, queries = synthetic_mixture(cfg["n"], cfg["dim"], seed=0xC0DD1E, device=device, style=cfg["style"],
                                          n_queries=cfg["n_queries"], latent_dim=cfg["latent_dim"])
        fi = build_index(base, cfg["branching"], cfg["leaf"], cfg["metric"], 20, seed=1, code_kind=cfg["code_kind"],
                         threshold="mean")
        idx = DeviceIndex(fi, 0)
    if os.environ.get("CUDDLE_PROBE_NQ"):  # a small request instead of the configuration's batch
        queries = queries[:int(os.environ["CUDDLE_PROBE_NQ"])].contiguous()
    nq, k = queries.shape[0], cfg["k"]
    ids = torch.empty((nq, k), dtype=torch.int64, device=device)
    dd = torch.empty((nq, k), dtype=torch.float32, device=device)
    cnt = torch.empty((nq,), dtype=torch.int32, device=device)
    for _ in range(3):
        idx.search_batch_device(queries, k, cfg["probes"], rf, 0, ids, dd, cnt)
    torch.cuda.synchronize()
    fn = _ffi.lib().cuddle_debug_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    fn.restype = ctypes.c_int
    out = np.zeros((16384, 8), dtype=np.uint64)
    assert fn(out.ctypes.data, 16384) == 0
    order = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(npts))  # stamp numbers in program order
    npts = len(order)
    t = out[:, order].astype(np.int64)
    t = t[(t[:, 0] > 0) & (t[:, npts - 1] > 0)]
    for i in range(1, npts):  # a phase the block's path skipped (no stamp) counts as zero time
        t[:, i] = np.where(t[:, i] > 0, t[:, i], t[:, i - 1])
    d = np.diff(t, axis=1) * 0.01  # 100 MHz ticks -> microseconds
    for i in range(npts - 1):
        print(f"stamp {order[i]} -> {order[i + 1]}: mean {d[:, i].mean():8.2f} us   p50 {np.median(d[:, i]):8.2f}   max {d[:, i].max():8.2f}")
    life = (t[:, -1] - t[:, 0]) * 0.01
    span = (t[:, -1].max() - t[:, 0].min()) * 0.01
    print(f"blocks {len(t)}; lifetime mean {life.mean():.2f} us; span {span:.1f} us; in flight (avg) {life.sum() / span:.1f}")


if __name__ == "__main__":
    main()
