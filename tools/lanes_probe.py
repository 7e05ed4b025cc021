import json, os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
import graph_replay_probe as g
import bench
name, rf = sys.argv[1], int(sys.argv[2])
cfg = bench.CONFIGS[name]
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
idx, queries, keep = g.build(name, dev)
nq, k, probes = queries.shape[0], cfg['k'], cfg['probes']
out = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
for _ in range(40): idx.search_batch_device(queries, k, probes, rf, 0, *out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): idx.search_batch_device(queries, k, probes, rf, 0, *out)
torch.cuda.synchronize()
print(json.dumps({'config': name, 'lanes_env': os.environ.get('CUDDLE_LANES'), 'ms': (time.perf_counter() - t0) / 50 * 1e3, 'checksum': int(out[0].sum())}))
