"""Is the ping-pong GEMM's epilogue bound per CU or chip-wide?  N = 3072, K = 768 (FFN1 / du shape, 12 K-tiles per output
tile), 256 x 256 tiles, mode 0: the full kernel and the build with DMA, MFMA and fragment reads compiled out (prologue +
barriers + epilogue), each with and without the epilogue (POLUS_GEMM_ABLATE bit 4), at grids of 24 .. 768 tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops

def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
N, K = 3072, 768
ops.set_env("POLUS_GEMM_PP", 256)
print(f"{'T':>6s} {'tiles':>6s} {'rounds':>6s} {'full us':>9s} {'no-epi us':>10s} {'epi-only us':>12s} {'neither us':>11s} {'noMFMA us':>10s} {'out MB':>7s}")
for T in (512, 2048, 4096, 5376, 10752, 16384):
    a, b = rnd(T, K), rnd(N, K) * 0.05
    c = torch.empty(T, N, dtype=dt, device=dev)
    fn = lambda: ops.gemm(a, b, c)
    res = {}
    for abl in (0, 16, 7, 23, 2):
        ops.set_env("POLUS_GEMM_ABLATE", abl)
        timed(fn, 3)
        res[abl] = min(timed(fn, 20) for _ in range(3))
    tiles = (T // 256) * (N // 256)
    mb = T * N * 2 / 1e6
    print(f"{T:6d} {tiles:6d} {tiles/256:6.2f} {res[0]*1e6:9.1f} {res[16]*1e6:10.1f} {res[7]*1e6:12.1f} {res[23]*1e6:11.1f} {res[2]*1e6:10.1f} {mb:7.1f}", flush=True)
ops.set_env("POLUS_GEMM_ABLATE"); ops.set_env("POLUS_GEMM_PP")
