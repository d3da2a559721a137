"""Grouped dW launch (the four weight gradients of an encoder layer) + its reduce: ring 256x128 vs ping-pong 256x256."""
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

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H, I = 768, 3072
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
probs = []
for n_out, n_in in shapes:
    probs.append((rnd(T, n_out), rnd(T, n_in), torch.empty(n_out, n_in, device=dev), torch.empty(n_out, device=dev)))
fl = sum(2.0 * T * o * i for o, i in shapes)
fn = lambda: ops.dense_bwd_params_grouped(probs, False, 0)
for name, env in (("ring, per-matrix reduce", {"POLUS_GEMM_PP": -1, "POLUS_DW_FUSED_REDUCE": 0}), ("ring, fused reduce", {"POLUS_GEMM_PP": -1}),
                  ("ping-pong, fused reduce", {}), ("ping-pong + stream-K remainder, delta 0", {"POLUS_DW_STREAMK": 1, "POLUS_DW_SK_DELTA": 0}),
                  ("... delta 2", {"POLUS_DW_STREAMK": 1, "POLUS_DW_SK_DELTA": 2}), ("... delta 4", {"POLUS_DW_STREAMK": 1, "POLUS_DW_SK_DELTA": 4}),
                  ("... delta 6", {"POLUS_DW_STREAMK": 1, "POLUS_DW_SK_DELTA": 6}), ("ping-pong, fused reduce (again)", {})):
    for k, v in env.items():
        ops.set_env(k, v)
    timed(fn, 3)
    t = min(timed(fn) for _ in range(5))
    for k in env:
        ops.set_env(k)
    print(f"{name:28s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s (GEMM + reduce)", flush=True)
