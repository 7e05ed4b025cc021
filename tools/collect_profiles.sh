#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics, the two HBM-traffic PMC passes
# and (MFMA=1) the matrix-core counter pass of one bench configuration, each a run of its own; leaves
# the (filtered) summaries in gpurun_out/profiles_<tag>/.
#   [MFMA=1] tools/collect_profiles.sh <config> <tag> [extra bench args]
set -e -o pipefail
cfg=$1; tag=$2; shift 2
root=$PWD
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
common="--config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/ks -o ks --output-format csv -- python3 "$root/bench.py" $common --steps 10 "$@" > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
cp /tmp/prof_$tag/ks/*kernel_stats.csv "$out/kernel_stats.csv"
keep() {  # search-path kernels only
  f=$(ls $1/*counter_collection.csv | head -1)
  head -1 "$f" > "$2"
  grep "cuddle::" "$f" | grep -v "assign_gemm\|assign_finish\|level_table\|level_sum\|level_mean" >> "$2" || true
}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c -d /tmp/prof_$tag/$c -o p --output-format csv -- python3 "$root/bench.py" $common --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  keep /tmp/prof_$tag/$c "$out/pmc_$c.csv"
done
cd "$root"
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE.csv" "$out/pmc_WRITE_SIZE.csv" "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py $common --steps 3 --warmup 1 $*" "$cfg" > "$out/pmc_hbm_traffic.json"
python3 tools/kstats.py "$out/kernel_stats.csv" > "$out/kernel_stats_summary.txt"
if [ -n "$MFMA" ]; then
  ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
  cd /tmp
  timeout -s KILL 300 rocprofv3 --pmc $ctr -d /tmp/prof_$tag/mfma -o p --output-format csv -- python3 "$root/bench.py" $common --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  keep /tmp/prof_$tag/mfma "$out/pmc_mfma.csv"
  cd "$root"
  python3 tools/pmc_mfma.py "$out/pmc_mfma.csv" "$out/kernel_stats.csv" "rocprofv3 --pmc $ctr --output-format csv -- python3 bench.py $common --steps 3 --warmup 1 $*" > "$out/pmc_mfma.json"
fi
