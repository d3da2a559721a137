"""Times polus_gemm on the Dense shapes of one BERT step (per-shape TFLOP/s), both kernels.

    python tools/gemm_bench.py [--T 16384] [--iters 20] [--large]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops  # noqa: E402


def bench(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=16384)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--pad", type=int, default=0, help="extra elements in every row stride (breaks power-of-two-ish strides)")
    args = ap.parse_args()
    H, I = (1024, 4096) if args.large else (768, 3072)
    T = args.T
    dt = torch.bfloat16
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    P = args.pad
    rnd = lambda r, c: (torch.rand(r, c + P, device=dev, generator=g) * 2 - 1).to(dt)[:, :c]
    shapes = [("qkv", T, 3 * H, H), ("out", T, H, H), ("ffn1", T, I, H), ("ffn2", T, H, I)]
    rows = []
    for name, M, N, K in shapes:
        if args.only and name not in args.only:
            continue
        a, b = rnd(M, K), rnd(N, K) * 0.05
        c = torch.empty(M, N + P, dtype=dt, device=dev)[:, :N]
        bias = torch.zeros(N, device=dev)
        aux = torch.empty(M, N + P, dtype=dt, device=dev)[:, :N]
        res = rnd(M, N)
        fl = 2.0 * M * N * K
        variants = {
            "plain": lambda: ops.gemm(a, b, c),
            "bias": lambda: ops.gemm(a, b, c, bias=bias),
            "bias+gelu+aux": lambda: ops.gemm(a, b, c, bias=bias, aux=aux, act="gelu", flags=ops.GEMM_ACT_FWD),
            "bias+resid": lambda: ops.gemm(a, b, c, bias=bias, resid=res),
            "bias+relu+aux": lambda: ops.gemm(a, b, c, bias=bias, aux=aux, act="relu", flags=ops.GEMM_ACT_FWD),
            "gelu-bwd(aux)": lambda: ops.gemm(a, b, c, aux=aux, act="gelu", flags=ops.GEMM_ACT_BWD),
            "relu-bwd(aux)": lambda: ops.gemm(a, b, c, aux=aux, act="relu", flags=ops.GEMM_ACT_BWD),
        }
        for vname, fn in variants.items():
            t3 = bench(fn, args.iters)
            ops.set_env("POLUS_GEMM_256", 1)
            t2 = bench(fn, args.iters)
            ops.set_env("POLUS_GEMM_256")
            ops.set_env("POLUS_GEMM_V1", 1)
            t1 = bench(fn, args.iters)
            ops.set_env("POLUS_GEMM_V1")
            rows.append((name, M, N, K, vname, fl / t3 / 1e12, t3 * 1e6, fl / t2 / 1e12, t2 * 1e6, fl / t1 / 1e12, t1 * 1e6))
        # dW shape: [N, K] = dY^T X, both K-strided, f32 out, split-K as the model uses it
        from polus_amd.layers import dw_split_k
        dy = rnd(M, N)
        gw = torch.empty(N, K, dtype=torch.float32, device=dev)
        sk = dw_split_k(N, K, M)
        t = bench(lambda: ops.gemm(dy, a, gw, a_layout=ops.K_STRIDED, b_layout=ops.K_STRIDED, split_k=sk), args.iters)
        rows.append((name, N, K, M, f"dW split_k={sk}", 0.0, 0.0, 0.0, 0.0, fl / t / 1e12, t * 1e6))
    print(f"{'gemm':6s} {'M':>6s} {'N':>6s} {'K':>6s} {'epilogue':16s} {'ring TF/s':>9s} {'us':>8s} {'256 TF/s':>9s} {'us':>8s} {'128 TF/s':>9s} {'us':>8s}")
    for r in rows:
        print(f"{r[0]:6s} {r[1]:6d} {r[2]:6d} {r[3]:6d} {r[4]:16s} {r[5]:9.1f} {r[6]:8.1f} {r[7]:9.1f} {r[8]:8.1f} {r[9]:9.1f} {r[10]:8.1f}")


if __name__ == "__main__":
    main()
