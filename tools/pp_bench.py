"""A/B of the K-contiguous bf16 GEMM kernels on the Dense shapes of one BERT step (forward and dX):
ring 256x128 (POLUS_GEMM_PP=-1) vs ping-pong 256x256 / 256x192, interleaved rounds in one process.

    python tools/pp_bench.py [--T 16384] [--rounds 5] [--iters 10] [--large]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--variants", default="-1,256,192")
    args = ap.parse_args()
    H, I = (1024, 4096) if args.large else (768, 3072)
    T = args.T
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
    # (name, N, K, epilogue)
    shapes = [("qkv fwd", 3 * H, H, "bias"), ("out fwd", H, H, "drop+resid"), ("ffn1 fwd", I, H, "gelu+aux"),
              ("ffn2 fwd", H, I, "drop+resid"), ("dx qkv", H, 3 * H, "resid"), ("dctx", H, H, "plain"),
              ("du (ffn2 dx)", I, H, "gelu-bwd"), ("da1 (ffn1 dx)", H, I, "resid")]
    sels = [int(v) for v in args.variants.split(",")]
    print(f"{'gemm':14s} {'N':>5s} {'K':>5s} {'epilogue':11s} " + " ".join(f"{('ring' if s < 0 else 'pp' + str(s)):>16s}" for s in sels))
    tot = {s: 0.0 for s in sels}
    totfl = 0.0
    for name, N, K, epi in shapes:
        a, b = rnd(T, K), rnd(N, K) * 0.05
        c = torch.empty(T, N, dtype=dt, device=dev)
        bias = torch.zeros(N, device=dev)
        aux = rnd(T, N)
        res = rnd(T, N)
        kw = {"bias": dict(bias=bias), "plain": dict(), "resid": dict(resid=res),
              "drop+resid": dict(bias=bias, resid=res, drop_p=0.1, seed=5),
              "gelu+aux": dict(bias=bias, aux=aux, act="gelu", flags=ops.GEMM_ACT_FWD),
              "gelu-bwd": dict(aux=aux, act="gelu", flags=ops.GEMM_ACT_BWD)}[epi]
        fn = lambda: ops.gemm(a, b, c, **kw)
        fl = 2.0 * T * N * K
        best = {s: 1e9 for s in sels}
        med = {s: [] for s in sels}
        for s in sels:
            ops.set_env("POLUS_GEMM_PP", s)
            timed(fn, 3)
        for _ in range(args.rounds):
            for s in sels:
                ops.set_env("POLUS_GEMM_PP", s)
                t = timed(fn, args.iters)
                best[s] = min(best[s], t)
                med[s].append(t)
        ops.set_env("POLUS_GEMM_PP")
        cells = []
        for s in sels:
            m = sorted(med[s])[len(med[s]) // 2]
            tot[s] += m
            cells.append(f"{fl / m / 1e12:7.1f} {m * 1e6:7.1f}us")
        totfl += fl
        print(f"{name:14s} {N:5d} {K:5d} {epi:11s} " + " ".join(f"{c_:>16s}" for c_ in cells), flush=True)
    print(f"{'all (median)':38s} " + " ".join(f"{totfl / tot[s] / 1e12:7.1f} {tot[s] * 1e6:7.1f}us" for s in sels))


if __name__ == "__main__":
    main()
