"""The persistent ping-pong GEMM issues its tile-queue atomic by inline asm and reads the returned register only behind an
explicit `s_waitcnt vmcnt(0)` much later (polus_amd/csrc/gemm_pp.hip).  hipcc does not know the register is pending, so this
checks the generated code of EVERY instantiation: between the atomic and the first wait for vmcnt(0) that follows it in the
listing, the destination register is neither read nor copied.  Exit code 1 on a violation."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "pp.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "polus_amd", "csrc", "gemm_pp.hip"), "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr); sys.exit(2)
    lines = open(out).read().splitlines()
bad = n = 0
func = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\S*gemm_pp[pm]_kernel\S*):", l)
    if m:
        func = m.group(1)
    m = re.search(r"global_atomic_add (v\d+), ", l)
    if not m or func is None:
        continue
    n += 1
    reg = m.group(1)
    j = i + 1
    while j < len(lines) and "s_endpgm" not in lines[j]:
        if re.search(r"\b%s\b" % reg, lines[j]):
            break
        j += 1
    # the first later mention of the register: a wait for vmcnt(0) must lie between (searching backwards from it)
    k = j - 1
    ok = False
    while k > i:
        if re.search(r"s_waitcnt vmcnt\(0\)", lines[k]):
            ok = True
            break
        if re.match(r"^\.LBB|^\s*s_cbranch|^\s*s_branch", lines[k]) and False:
            pass
        k -= 1
    # control flow: the mention must be in straight-line code behind that wait (no label between the wait and the mention)
    if ok and any(re.match(r"^\.LBB", x) for x in lines[k:j]):
        ok = False
    print(f"{func[:60]:60s} {reg:5s} atomic @{i}  first use @{j}  {'ok' if ok else 'NOT behind a vmcnt(0) wait'}")
    bad += 0 if ok else 1
print(f"{n} tile-queue atomics checked, {bad} bad")
sys.exit(1 if bad or n == 0 else 0)
