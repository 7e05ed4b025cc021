#!/bin/bash
# tools/scan_sweep.sh <config> <rerank> lib[:threads] ... : scan-stage time of bench.py for several builds of the library
cfg=$1; rf=$2; shift 2
o=gpurun_out/sweep; mkdir -p $o
for v in "$@"; do
  lib=${v%%:*}; thr=${v#*:}; [ "$thr" = "$v" ] && thr=""
  export CUDDLE_HIP_LIB=$PWD/vectordb_amd/libcuddle_$lib.so
  if [ -n "$thr" ]; then export CUDDLE_SCAN_THREADS=$thr; else unset CUDDLE_SCAN_THREADS; fi
  timeout -k 10 300 python -u bench.py --config $cfg --rerank $rf --no-cpu-baseline --no-hbm-probe --steps 20 > $o/$cfg.$lib.$thr.json 2> $o/$cfg.$lib.$thr.err || { tail -5 $o/$cfg.$lib.$thr.err; exit 1; }
  python - "$o/$cfg.$lib.$thr.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(f"{sys.argv[2]:14s} qps {d['value']/1e6:6.3f}M  scan {r['avg_launch_ms']*1e3:7.1f} us  {r['achieved']:7.0f} GB/s  frac {r['frac']:.3f}  stages {' '.join(f'{k}={v*1e3:.0f}' for k,v in r['stage_ms_per_step'].items())}  recall {[v for k,v in d.items() if k.startswith('recall_at_')][0]:.3f}")
PY
done
