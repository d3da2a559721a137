"""Summarise a rocprofv3 *_kernel_stats.csv per training step: python tools/kstats.py <csv> <n_steps>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "")
    n = re.sub(r"^_ZN\d+_GLOBAL__N_1\d*", "", n)[:64]
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if ms > 0.02:
        print(f"{n:64s} calls/step {int(r['Calls']) / steps:6.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {ms:7.3f}")
print(f"total kernel time per step: {tot:.3f} ms")
