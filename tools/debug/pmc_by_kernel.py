"""Mean of every counter per kernel name from a rocprofv3 --pmc output directory (counter_collection.csv)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(t in k for t in sys.argv[2:] or [""]):
        continue
    print(k)
    base = None
    for name, vals in sorted(cs.items()):
        m = sum(vals) / len(vals)
        if name == "SQ_WAVE_CYCLES": base = m
    for name, vals in sorted(cs.items()):
        m = sum(vals) / len(vals)
        print(f"    {name:28s} {m:16.0f}" + (f"   {m / base:6.3f} of SQ_WAVE_CYCLES" if base and name.startswith("SQ_") and "BUSY" not in name else ""))
