"""Does running the CUs out of phase pay?  FFN1 / du shape (T = 16384, N = 3072, K = 768; 768 tiles of 256 x 256 = 3 per
CU).  POLUS_GEMM_ABLATE = 64 + (delay << 8): every other first-round workgroup starts `delay` x 10 ns late, so half of the
CUs are in their epilogues while the others are in their K loops.  Reported: the time, and the time less the injected
delay (what a stagger obtained for free -- e.g. from uneven first tiles -- would give).  HIP-graph replay of 20 launches, so
that the host's launch rate does not floor the numbers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from polus_amd._lib import GEMM_ACT_FWD, GEMM_ACT_BWD
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
T, N, K = 16384, 3072, 768
a, b = rnd(T, K), rnd(N, K) * 0.05
bias = torch.zeros(N, device=dev)
c = torch.empty(T, N, dtype=dt, device=dev); aux = torch.empty(T, N, dtype=dt, device=dev); u = rnd(T, N)
cases = [("mode 0: bias only", lambda: ops.gemm(a, b, c, bias=bias)),
         ("mode 1: gelu + aux store (FFN1)", lambda: ops.gemm(a, b, c, bias=bias, act="gelu", aux=aux, flags=GEMM_ACT_FWD)),
         ("mode 3: gelu' + aux load (du)", lambda: ops.gemm(a, b, c, act="gelu", aux=u, flags=GEMM_ACT_BWD))]

def graph_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best

ops.set_env("POLUS_GEMM_PP", 256)
delays_us = (0, 2, 4, 6, 8, 10, 12)
print(f"{'':34s} {'baseline':>9s} " + " ".join(f"{'d=' + str(d) + 'us':>14s}" for d in delays_us))
for name, fn in cases:
    ops.set_env("POLUS_GEMM_ABLATE", 0)
    base = graph_time(fn)
    cells = []
    for d in delays_us:
        ops.set_env("POLUS_GEMM_ABLATE", 64 + ((d * 100) << 8))
        t = graph_time(fn)
        cells.append(f"{t*1e6:6.1f} ({t*1e6 - d:5.1f})")
    print(f"{name:34s} {base*1e6:7.1f}us " + " ".join(f"{x:>14s}" for x in cells), flush=True)
ops.set_env("POLUS_GEMM_ABLATE"); ops.set_env("POLUS_GEMM_PP")
