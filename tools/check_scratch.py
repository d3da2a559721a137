"""Fails if a hot kernel uses scratch memory (a silent 5x slowdown: spilled MFMA accumulators).
    python tools/check_scratch.py            # compiles device code of the hot files to .s and greps
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ["gemm_p.hip", "gemm_ring.hip", "gemm256.hip", "gemm.hip", "attention.hip"]
bad = 0
with tempfile.TemporaryDirectory() as td:
    for f in HOT:
        out = os.path.join(td, f + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                            os.path.join(ROOT, "polus_amd", "csrc", f), "-o", out], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr); sys.exit(2)
        txt = open(out).read()
        n_scratch, n_flat = txt.count("scratch_"), txt.count("flat_load")
        print(f"{f:16s} scratch ops {n_scratch:4d}   flat loads {n_flat:4d}")
        bad += n_scratch
sys.exit(1 if bad else 0)
