#!/bin/bash
# tools/ktrace.sh <config> <tag> [bench args]: rocprofv3 kernel trace of one bench configuration, timeline of its last search
cfg=$1; tag=$2; shift 2
root=$PWD; out=$root/gpurun_out/ktrace_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kt_$tag
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o kt --output-format csv -- python3 $root/bench.py --config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold --steps 8 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(ls /tmp/kt_$tag/*kernel_trace.csv /tmp/kt_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
cd $root; python3 tools/step_timeline.py $f > $out/timeline.txt; cat $out/timeline.txt
