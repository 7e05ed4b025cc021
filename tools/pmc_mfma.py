"""Matrix-core utilisation of the library's kernels that issue MFMA instructions, from one rocprofv3
--pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA
SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE) and the kernel-trace statistics of the same command.

    python tools/pmc_mfma.py counter_collection.csv kernel_stats.csv "<command>" > profiles/r02_<config>_pmc_mfma.json

Per kernel and launch: MFMA instructions, bf16 + f16 MFMA flops (MOPS x 512; the two types share a peak), the share of the kernel's
SIMD-cycles in which the MFMA pipe was busy (busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs):
rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs), and flops / average duration against the 2.5 PFLOP/s
dense bf16 peak."""
import collections
import csv
import json
import re
import sys

SIMDS, XCDS, PEAK_BF16 = 1024, 8, 2.5e15


def base_name(kernel):
    k = kernel.split("(")[0]
    k = re.sub(r"<.*", "", k)
    return k.split("::")[-1].replace("void ", "").strip()


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(sys.argv[1])):
        if "cuddle::" not in r["Kernel_Name"]:
            continue
        k = base_name(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    dur = {}
    for r in csv.DictReader(open(sys.argv[2])):
        if "cuddle::" in r["Name"]:
            k = base_name(r["Name"])
            calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
            c0, t0 = dur.get(k, (0, 0.0))
            dur[k] = (c0 + calls, t0 + total)
    out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "peak_bf16_dense_flops": PEAK_BF16,
           "note": "per launch; mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)", "kernels": {}}
    for k, v in sorted(acc.items()):
        n = {c: max(cnt[k][c], 1) for c in v}
        mean = {c: v[c] / n[c] for c in v}
        if mean.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        flops = (mean.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + mean.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)) * 512
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        e = {"launches": n.get("SQ_INSTS_MFMA", 0), "counters_mean": {c: round(x, 1) for c, x in sorted(mean.items())},
             "mfma_flops_bf16_f16": flops,
             "mfma_busy_share": (mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS)) if gui else None}
        if k in dur and dur[k][0]:
            avg_ns = dur[k][1] / dur[k][0]
            e["avg_duration_us_kernel_trace"] = avg_ns / 1e3
            e["achieved_tflops"] = flops / avg_ns / 1e3
            e["frac_of_dense_bf16_peak"] = flops / (avg_ns * 1e-9) / PEAK_BF16
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
