#!/bin/bash
# tools/pmc_valu.sh <config> <kernel name pattern> <tag>: vector-unit counters of one kernel of a bench configuration
# (one rocprofv3 --pmc pass of its own) -> gpurun_out/pmc_valu_<tag>.txt
set -e -o pipefail
cfg=$1; pat=$2; tag=$3
root=$PWD; raw=/tmp/pv_$tag
export TMPDIR=/tmp; cd /tmp; rm -rf $raw
ctr="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 300 rocprofv3 --pmc $ctr -d $raw -o p --output-format csv -- python3 "$root/bench.py" --config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold --steps 3 --warmup 1 > /dev/null 2> $raw.err || { tail -5 $raw.err; exit 1; }
f=$(find $raw -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter file"; tail -20 $raw.err; exit 1; }
cd "$root"
{ echo "# rocprofv3 --pmc $ctr -- python3 bench.py --config $cfg --no-cpu-baseline --no-hbm-probe --no-reference-threshold --steps 3 --warmup 1"; echo "# per launch, summed over the device (SQ_* counters count per SE/XCD sample as rocprofv3 reports them)"; python3 tools/pmc_kernel.py "$f" "$pat"; } > gpurun_out/pmc_valu_$tag.txt
rm -rf $raw
