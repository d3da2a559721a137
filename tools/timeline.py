"""One training step as a timeline, from a rocprofv3 --kernel-trace CSV (run with the side streams ON):
start offset, duration, queue and name of every kernel of the LAST complete step, the gaps in which no kernel runs, and how long
each queue's kernels overlap kernels of other queues.

    python tools/timeline.py <dir with *_kernel_trace.csv> [marker kernel substring, default adam_kernel... see below]
A step is delimited by the embed_fwd kernel (first kernel of the forward pass)."""
import csv
import glob
import re
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "embed_fwd" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0]["s"]
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:44]
qs = sorted({r["Queue_Id"] for r in step})
print(f"# {f}\n# step: {len(step)} kernels, {(rows[b]['s'] - t0) / 1e3:.1f} us start to start; queues {qs}")
end_max = t0
idle = 0.0
for r in step:
    gap = (r["s"] - end_max) / 1e3
    if gap > 0:
        idle += gap
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{qs.index(r['Queue_Id'])} {'gap %6.1f' % gap if gap > 1.0 else '          '} {short(r['Kernel_Name'])}")
    end_max = max(end_max, r["e"])
print(f"# no kernel running for {idle:.1f} us of the step")
