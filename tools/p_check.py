"""Persistent 256x192 GEMM (gemm_p.hip) against the ring kernel and an f32 torch reference,
every epilogue variant, interior and ragged shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)
bad = 0
for (M, N, K) in [(256, 192, 64), (512, 384, 128), (16384, 768, 768), (2048, 2304, 768), (300, 400, 192), (1000, 192, 3072), (16384, 3072, 768)]:
    a = rnd(M, K).bfloat16(); b = (rnd(N, K) * 0.1).bfloat16()
    bias = rnd(N).float(); res = rnd(M, N).bfloat16(); u = rnd(M, N).bfloat16()
    ref = a.float() @ b.float().t()
    variants = {
        "plain": (dict(), lambda r: r),
        "bias": (dict(bias=bias), lambda r: r + bias),
        "bias+gelu+aux": (dict(bias=bias, aux="new", act="gelu", flags=ops.GEMM_ACT_FWD), lambda r: torch.nn.functional.gelu(r + bias)),
        "bias+resid": (dict(bias=bias, resid=res), lambda r: r + bias + res.float()),
        "gelu-bwd": (dict(aux=u, act="gelu", flags=ops.GEMM_ACT_BWD), None),
        "drop+resid": (dict(bias=bias, resid=res, drop_p=0.1, seed=77), None),
        "f32-accum": (dict(flags=ops.GEMM_ACCUM_C, f32=True), None),
    }
    for name, (kw, fn) in variants.items():
        outs = []
        for mode in ("0", "2"):
            ops.set_env("POLUS_GEMM_P", mode)
            kw2 = dict(kw)
            f32 = kw2.pop("f32", False)
            c = torch.full((M, N), 0.5, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
            if kw2.get("aux") == "new":
                kw2["aux"] = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm(a, b, c, **kw2)
            torch.cuda.synchronize()
            outs.append((c.float(), kw2.get("aux").float() if name == "bias+gelu+aux" else None))
        d = (outs[0][0] - outs[1][0]).abs().max().item()
        scale = outs[0][0].abs().max().item()
        msg = f"{M}x{N}x{K} {name:14s} p-vs-ring max|d| {d:.3e} (scale {scale:.2f})"
        tol = 2e-2 * max(scale, 1.0)
        if name == "drop+resid":
            # identical mask (counter-based): differences only from rounding
            pass
        if fn is not None:
            e = (outs[1][0] - fn(ref)).abs().max().item()
            msg += f"  p-vs-f32ref {e:.3e}"
            if e > tol: bad += 1; msg += "  <-- BAD"
        if outs[0][1] is not None:
            da = (outs[0][1] - outs[1][1]).abs().max().item(); msg += f" aux d {da:.2e}"
            if da > tol: bad += 1; msg += " <-- BAD aux"
        if d > tol: bad += 1; msg += "  <-- BAD"
        print(msg, flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
