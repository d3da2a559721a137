"""Per-kernel mean of rocprofv3 counter_collection.csv counters: python tools/pmc_summary.py <csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
