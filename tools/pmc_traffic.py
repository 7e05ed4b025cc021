"""HBM traffic per launch of the library's kernels from two rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE collected in separate runs of the same command): reads = FETCH_SIZE KB x 1024 x 2 (the
gfx950 correction for wide coalesced reads), writes = WRITE_SIZE KB x 1024. Writes the JSON that
bench.py reads for roofline.traffic (profiles/pmc_hbm_traffic_<config>.json).

    python tools/pmc_traffic.py FETCH.csv WRITE.csv "<command>" "<workload>" > profiles/pmc_hbm_traffic_gist1m.json
"""
import collections
import csv
import json
import re
import sys


def base_name(kernel):
    k = kernel.split("(")[0]
    k = re.sub(r"<.*", "", k)
    return k.split("::")[-1].replace("void ", "").strip()


def per_kernel(path, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if "cuddle::" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        k = base_name(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "workload": sys.argv[4] if len(sys.argv) > 4 else "",
           "correction": "FETCH_SIZE (KB) x 1024 x 2 on gfx950 for wide coalesced reads; WRITE_SIZE (KB) x 1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        rd, wr = f * 1024 * 2, w * 1024
        out["kernels"][k] = {"FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "launches": nf,
                             "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
                             "traffic_bytes_per_launch": rd + wr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
