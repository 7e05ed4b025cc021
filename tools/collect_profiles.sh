#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics plus the two HBM-traffic PMC
# passes of one bench configuration; leaves the (filtered) summaries in gpurun_out/profiles_<tag>/.
#   tools/collect_profiles.sh <config> <tag> [extra bench args]
set -e -o pipefail
cfg=$1; tag=$2; shift 2
root=$PWD
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/ks -o ks --output-format csv -- python3 "$root/bench.py" --config "$cfg" --no-cpu-baseline --steps 10 "$@" > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
cp /tmp/prof_$tag/ks/*kernel_stats.csv "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c -d /tmp/prof_$tag/$c -o p --output-format csv -- python3 "$root/bench.py" --config "$cfg" --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  f=$(ls /tmp/prof_$tag/$c/*counter_collection.csv | head -1)
  head -1 "$f" > "$out/pmc_$c.csv"
  grep "cuddle::" "$f" | grep -v "assign_gemm\|assign_finish\|level_table\|level_sum\|level_mean" >> "$out/pmc_$c.csv" || true  # search-path kernels only
done
cd "$root"
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE.csv" "$out/pmc_WRITE_SIZE.csv" "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --config $cfg --no-cpu-baseline --steps 3 --warmup 1 $*" "$cfg" > "$out/pmc_hbm_traffic.json"
python3 tools/kstats.py "$out/kernel_stats.csv" > "$out/kernel_stats_summary.txt"
