"""Print the per-kernel averages of a rocprofv3 kernel_stats.csv (library kernels only)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "cuddle" in r["Name"]:
        print(f"{r['Name'].split('(')[0][-58:]:58s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1000:9.1f} us")
