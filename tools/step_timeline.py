"""Timeline of one search from a rocprofv3 kernel trace: every kernel between the last two launches of the
query quantiser, with its duration and the idle gap before it.
usage: python tools/step_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if any(n in r["Kernel_Name"] for n in ("quantize_init_kernel", "binary_quantize_kernel", "rabitq_query_prep"))]
a, b = first[-2], first[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cuddle::", "")[:60]
    print(f"{(s - t0) / 1000:9.1f} us  gap {(s - prev_end) / 1000:6.1f}  dur {(e - s) / 1000:7.1f}  grid {r.get('Grid_Size', '?'):>9s}  {name}")
    prev_end = e
print(f"total {(prev_end - t0) / 1000:.1f} us")
