"""Ablations of the ping-pong GEMM main loop (POLUS_GEMM_ABLATE bits: 1 no in-loop DMA, 2 no MFMA, 4 no fragment reads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops

def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

T = 16384
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
print(f"{'shape':22s} " + " ".join(f"{n:>12s}" for n in ["full", "noDMA", "noMFMA", "noDMA+MFMA", "noRead", "noDMA+Read", "noMFMA+Read", "barriers", "M-prio"]))
for tn, N, K in [(256, 3072, 768), (256, 768, 3072), (192, 768, 3072), (192, 2304, 768), (192, 768, 768)]:
    a, b = rnd(T, K), rnd(N, K) * 0.05
    c = torch.empty(T, N, dtype=dt, device=dev)
    fn = lambda: ops.gemm(a, b, c)
    ops.set_env("POLUS_GEMM_PP", tn)
    cells = []
    for abl in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        ops.set_env("POLUS_GEMM_ABLATE", abl)
        timed(fn, 3)
        t = min(timed(fn) for _ in range(3))
        cells.append(f"{t*1e6:7.1f}us")
    ops.set_env("POLUS_GEMM_ABLATE"); ops.set_env("POLUS_GEMM_PP")
    print(f"pp{tn} N={N:5d} K={K:5d}  " + " ".join(f"{c_:>12s}" for c_ in cells), flush=True)
