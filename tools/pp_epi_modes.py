"""What the GELU epilogues of the ping-pong GEMM cost at the FFN1 / du shape (T = 16384, N = 3072, K = 768, 256 x 256 tiles),
by switching their parts off through the runtime arguments: activation none vs gelu, with / without the pre-activation (aux)
store, plain bias epilogue for reference.  Isolated launches on warm operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from polus_amd._lib import GEMM_ACT_FWD, GEMM_ACT_BWD

def timed(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
T, N, K = 16384, 3072, 768
a, b = rnd(T, K), rnd(N, K) * 0.05
bias = torch.zeros(N, device=dev)
c = torch.empty(T, N, dtype=dt, device=dev); aux = torch.empty(T, N, dtype=dt, device=dev)
u = rnd(T, N)
cases = [
    ("mode 0: bias only",                         lambda: ops.gemm(a, b, c, bias=bias)),
    ("mode 1: act none, no aux store",            lambda: ops.gemm(a, b, c, bias=bias, act=None, flags=GEMM_ACT_FWD)),
    ("mode 1: act none, aux store",               lambda: ops.gemm(a, b, c, bias=bias, act=None, aux=aux, flags=GEMM_ACT_FWD)),
    ("mode 1: gelu, no aux store",                lambda: ops.gemm(a, b, c, bias=bias, act="gelu", flags=GEMM_ACT_FWD)),
    ("mode 1: gelu, aux store (FFN1)",            lambda: ops.gemm(a, b, c, bias=bias, act="gelu", aux=aux, flags=GEMM_ACT_FWD)),
    ("mode 3: act none, aux load",                lambda: ops.gemm(a, b, c, act=None, aux=u, flags=GEMM_ACT_BWD)),
    ("mode 3: gelu', aux load (du)",              lambda: ops.gemm(a, b, c, act="gelu", aux=u, flags=GEMM_ACT_BWD)),
]
resid = rnd(T, N)
cases += [
    ("mode 2: residual",                          lambda: ops.gemm(a, b, c, bias=bias, resid=resid)),
    ("mode 2: dropout 0.1 + residual",            lambda: ops.gemm(a, b, c, bias=bias, resid=resid, drop_p=0.1, seed=5)),
]
for tn in (256, 192):
    ops.set_env("POLUS_GEMM_PP", tn)
    print(f"-- 256 x {tn} tiles")
    for name, fn in cases:
        timed(fn, 3)
        t = min(timed(fn) for _ in range(3))
        print(f"{name:40s} {t*1e6:7.1f} us  {2*T*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
ops.set_env("POLUS_GEMM_PP")
