"""Experiment: the four dW GEMMs of a layer one after another with chip-filling split-K (what the
model does) against the same four run concurrently on four streams with few splits (an
approximation of one grouped launch): GEMM + split-K reduce time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from polus_amd.layers import dw_split_k
T = 16384
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).bfloat16()
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
dys = [rnd(T, n) for n, k in shapes]; xs = [rnd(T, k) for n, k in shapes]
gws = [torch.empty(n, k, dtype=torch.float32, device=dev) for n, k in shapes]
gbs = [torch.empty(n, dtype=torch.float32, device=dev) for n, k in shapes]
streams = [torch.cuda.Stream() for _ in shapes]

def seq(splits):
    for (n, k), dy, x, gw, gb, sk in zip(shapes, dys, xs, gws, gbs, splits):
        ops.dense_bwd_params(dy, x, gw, gb, split_k=sk)

def conc(splits):
    cur = torch.cuda.current_stream()
    for st in streams: st.wait_stream(cur)
    for (n, k), dy, x, gw, gb, sk, st in zip(shapes, dys, xs, gws, gbs, splits, streams):
        with torch.cuda.stream(st):
            ops.dense_bwd_params(dy, x, gw, gb, split_k=sk)
    for st in streams: cur.wait_stream(st)

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

fl = sum(2.0 * T * n * k for n, k in shapes)
base = [dw_split_k(n, k, T) for n, k in shapes]
t = bench(lambda: seq(base)); print(f"sequential, splits {base}: {t:.1f} us  {fl/t/1e6:.0f} TF")
for sp in ([2, 2, 2, 2], [2, 4, 2, 2], [3, 6, 2, 2], [1, 2, 1, 1], [4, 8, 3, 3]):
    t = bench(lambda: conc(sp)); print(f"concurrent, splits {sp}: {t:.1f} us  {fl/t/1e6:.0f} TF")
    t = bench(lambda: seq(sp)); print(f"sequential, splits {sp}: {t:.1f} us  {fl/t/1e6:.0f} TF")
