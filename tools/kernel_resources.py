"""Registers / LDS / occupancy of every kernel of one source file, from hipcc -Rpass-analysis=kernel-resource-usage:
    python tools/kernel_resources.py gemm_pp [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "polus_amd", "csrc", sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", src, "-o", os.path.join(td, "x.o")], capture_output=True, text=True)
cur = {}
def flush():
    if cur and flt in cur.get("name", ""):
        print(f"{cur['name'][:70]:70s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>4s} sgpr {cur.get('TotalSGPRs','?'):>4s} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} lds {cur.get('LDS Size [bytes/block]','?')}")
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        flush(); cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")}
    else:
        cur[k] = v
flush()
