#!/bin/bash
# Run on the GPU box from the repository root: the GPU test suite, then the bench line of every
# configuration (gist1m and sift1m with the CPU baseline), results under gpurun_out/final_<tag>_*.
#   tools/final_numbers.sh <tag>
set -e -o pipefail
tag=$1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -20 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for c in gist1m sift1m; do timeout -k 10 400 python bench.py --config $c > gpurun_out/final_${tag}_$c.json 2> gpurun_out/final_${tag}_$c.err; done
for c in shard12m shard125m rabitq10m gist10m; do timeout -k 10 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/final_${tag}_$c.json 2> gpurun_out/final_${tag}_$c.err; done
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for c in ("gist1m", "sift1m", "shard12m", "shard125m", "rabitq10m", "gist10m"):
    j = json.loads(open(f"gpurun_out/final_{tag}_{c}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    cb = j.get("cpu_baseline") or {}
    print(c, round(j["value"]), round(j["ms_per_step"], 3), "p99", round(j["p99_batch_ms"], 3),
          [v for k, v in j.items() if k.startswith("recall_at")], {k: round(v, 3) for k, v in r["stage_ms_per_step"].items()},
          round(r["achieved"]), round(r["frac"], 3), cb.get("value"), (cb.get("reference_shaped") or {}).get("value"),
          cb.get("gpu_ids_identical_on_sample"), j["route_fallbacks_per_batch"])
PY
