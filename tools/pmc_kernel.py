"""Average the counters of a rocprofv3 --pmc counter_collection.csv per kernel (per launch)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
pat = sys.argv[2] if len(sys.argv) > 2 else "cuddle"
first = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-56:]
    if pat not in r["Kernel_Name"]:
        continue
    first = first or r["Counter_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first:
        cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "launches", n, {c: round(x / n) for c, x in sorted(v.items())})
