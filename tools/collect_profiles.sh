#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics, the two HBM-traffic PMC passes
# and (MFMA=1) the matrix-core counter pass of one bench configuration, each a run of its own; leaves
# the summaries (not the raw counter files) in gpurun_out/profiles_<tag>/.
#   [MFMA=1] tools/collect_profiles.sh <config> <tag> [extra bench args]
set -e -o pipefail
cfg=$1; tag=$2; shift 2
root=$PWD
out=$root/gpurun_out/profiles_$tag
raw=/tmp/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf $raw
common="--config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $raw/ks -o ks --output-format csv -- python3 "$root/bench.py" $common --steps 10 "$@" > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
grep "Name\|cuddle::" $raw/ks/*kernel_stats.csv > "$out/kernel_stats.csv"
keep() {  # search-path kernels only
  f=$(ls $1/*counter_collection.csv | head -1)
  head -1 "$f" > "$2"
  grep "cuddle::" "$f" | grep -v "assign_gemm\|assign_finish\|level_table\|level_sum\|level_mean\|kmeans\|rabitq_encode" >> "$2" || true
}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c -d $raw/$c -o p --output-format csv -- python3 "$root/bench.py" $common --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  keep $raw/$c $raw/pmc_$c.csv
done
cd "$root"
python3 tools/pmc_traffic.py $raw/pmc_FETCH_SIZE.csv $raw/pmc_WRITE_SIZE.csv "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py $common --steps 3 --warmup 1 $*" "$cfg" > "$out/pmc_hbm_traffic.json"
python3 tools/kstats.py "$out/kernel_stats.csv" > "$out/kernel_stats_summary.txt"
if [ -n "$MFMA" ]; then
  ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
  cd /tmp
  timeout -s KILL 300 rocprofv3 --pmc $ctr -d $raw/mfma -o p --output-format csv -- python3 "$root/bench.py" $common --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  keep $raw/mfma $raw/pmc_mfma.csv
  cd "$root"
  python3 tools/pmc_mfma.py $raw/pmc_mfma.csv "$out/kernel_stats.csv" "rocprofv3 --pmc $ctr --output-format csv -- python3 bench.py $common --steps 3 --warmup 1 $*" > "$out/pmc_mfma.json"
fi
rm -rf $raw
