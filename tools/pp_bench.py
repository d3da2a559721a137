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


def timed_cold(fn, iters, scrub):
    """Events around each launch alone; a 512 MiB pass in between evicts the operands from L2 and the Infinity Cache."""
    pairs = []
    for _ in range(iters):
        scrub.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=16384)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--variants", default="-1,256,192")
    ap.add_argument("--ab", default=None, help="NAME=v1,v2,..: A/B another library switch (POLUS_GEMM_PP stays on its per-shape choice)")
    ap.add_argument("--only", default=None, help="comma-separated substrings: only the shapes whose name contains one")
    ap.add_argument("--cold", action="store_true", help="touch 512 MiB between launches (operands no longer L2 / Infinity-Cache warm)")
    args = ap.parse_args()
    env_name = "POLUS_GEMM_PP"
    if args.ab:
        env_name, vals = args.ab.split("=")
        args.variants = vals
    H, I = (1024, 4096) if args.large else (768, 3072)
    T = args.T
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
    # (name, N, K, epilogue)
    shapes = [("qkv fwd", 3 * H, H, "bias"), ("out fwd", H, H, "drop+resid"), ("ffn1 fwd", I, H, "gelu+aux"), ("ffn1 noaux", I, H, "gelu"),
              ("ffn2 fwd", H, I, "drop+resid"), ("dx qkv", H, 3 * H, "resid"), ("dctx", H, H, "plain"),
              ("du (ffn2 dx)", I, H, "gelu-bwd"), ("da1 (ffn1 dx)", H, I, "resid")]
    if args.only:
        shapes = [sh for sh in shapes if any(o in sh[0] for o in args.only.split(","))]
    sels = [int(v) for v in args.variants.split(",")]
    label = (lambda s: 'ring' if s < 0 else 'pp' + str(s)) if not args.ab else (lambda s: f"{env_name[6:]}={s}")
    print(f"{'gemm':14s} {'N':>5s} {'K':>5s} {'epilogue':11s} " + " ".join(f"{label(s):>16s}" for s in sels))
    scrub = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if args.cold else None
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
              "gelu": dict(bias=bias, act="gelu", flags=ops.GEMM_ACT_FWD),
              "gelu-bwd": dict(aux=aux, act="gelu", flags=ops.GEMM_ACT_BWD)}[epi]
        fn = lambda: ops.gemm(a, b, c, **kw)
        fl = 2.0 * T * N * K
        best = {s: 1e9 for s in sels}
        med = {s: [] for s in sels}
        for s in sels:
            ops.set_env(env_name, s)
            timed(fn, 3)
        for _ in range(args.rounds):
            for s in sels:
                ops.set_env(env_name, s)
                t = timed_cold(fn, args.iters, scrub) if args.cold else timed(fn, args.iters)
                best[s] = min(best[s], t)
                med[s].append(t)
        ops.set_env(env_name)
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
