#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics of one bench run, summary only.
#   tools/kprof.sh <tag> [bench args]     -> gpurun_out/kprof_<tag>.txt, gpurun_out/kprof_<tag>.json
set -e -o pipefail
tag=$1; shift
root=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kprof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kprof_$tag -o ks --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --steps 10 --warmup 3 "$@" > "$root/gpurun_out/kprof_$tag.json" 2> "$root/gpurun_out/kprof_$tag.err"
cd "$root"
python3 tools/kstats.py /tmp/kprof_$tag/*kernel_stats.csv > gpurun_out/kprof_$tag.txt
