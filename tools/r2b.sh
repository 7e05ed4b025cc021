#!/bin/bash
# one GPU-box call: no-alloc test alone, 2-rank rehearsals (gloo on one GPU), default bench, scan phase times
set -o pipefail
o=gpurun_out/r2b; mkdir -p $o
timeout -k 10 200 python -u -m pytest tests/test_zz_no_alloc.py -q --timeout 150 --timeout-method thread > $o/noalloc_alone.log 2>&1 || { tail -5 $o/noalloc_alone.log; exit 1; }
tail -1 $o/noalloc_alone.log
CUDDLE_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --config tinytree --steps 5 --warmup 2 > $o/tinytree_2rank.json 2> $o/tinytree_2rank.err || { tail -20 $o/tinytree_2rank.err; exit 1; }
echo "tinytree x2: $(cut -c1-300 $o/tinytree_2rank.json)"
CUDDLE_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --config tiny --steps 5 --warmup 2 > $o/tiny_2rank.json 2> $o/tiny_2rank.err || { tail -20 $o/tiny_2rank.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2b/tiny_2rank.json"))
print("tiny x2: n_gpus", d["n_gpus"], "recall", d["recall_at_10"], "identical to oracle:", d["cpu_baseline"]["gpu_ids_identical_on_sample"])
PY
timeout -k 10 600 python -u bench.py > $o/bench_default.json 2> $o/bench_default.err || { tail -20 $o/bench_default.err; exit 1; }
tail -12 $o/bench_default.err
CUDDLE_HIP_LIB=$PWD/vectordb_amd/libcuddle_timing.so timeout -k 10 300 python -u tools/kernel_phase_times.py shard12m 25 5 > $o/phase_shard12m.txt 2>&1 || { tail -20 $o/phase_shard12m.txt; exit 1; }
cat $o/phase_shard12m.txt
