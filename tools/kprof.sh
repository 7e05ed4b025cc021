#!/bin/bash
# tools/kprof.sh <config> <tag> [bench args]: rocprofv3 kernel statistics of one bench configuration -> gpurun_out/kprof_<tag>/
cfg=$1; tag=$2; shift 2
root=$PWD; out=$root/gpurun_out/kprof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kp_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kp_$tag -o ks --output-format csv -- python3 $root/bench.py --config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold --steps 10 "$@" > $out/bench_under_rocprof.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cp /tmp/kp_$tag/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || cp /tmp/kp_$tag/*/*kernel_stats.csv $out/kernel_stats.csv
cd $root; python3 tools/kstats.py $out/kernel_stats.csv > $out/kernel_stats_summary.txt; head -40 $out/kernel_stats_summary.txt
