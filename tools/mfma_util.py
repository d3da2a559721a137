"""MFMA utilisation per kernel from a rocprofv3 --pmc pass holding SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
busy cycles summed over the chip's SIMDs / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
counter is summed over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back).  Also the wave-cycle split when present."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    m = {n: sum(v) / len(v) for n, v in cs.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    if m["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
        continue
    util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    extra = ""
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        w = m["SQ_WAVE_CYCLES"]
        extra = "  wave cycles: active %.2f  issue-stall %.2f  parked %.2f" % (m.get("SQ_ACTIVE_INST_ANY", 0) / w, m.get("SQ_WAIT_INST_ANY", 0) / w, m.get("SQ_WAIT_ANY", 0) / w)
    rows.append((len(cs["GRBM_GUI_ACTIVE"]) * cyc, f"{k[:78]:78s} launches {len(cs['GRBM_GUI_ACTIVE']):4d}  kernel cycles {cyc:9.0f}  MFMA busy {util:5.3f}{extra}"))
for _, line in sorted(rows, reverse=True):
    print(line)
