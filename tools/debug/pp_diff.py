import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(512, 768, 768), (256, 192, 64), (256, 192, 128), (256, 384, 192)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    b = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1).to(torch.bfloat16)
    ref = a.float() @ b.float().T
    outs = {}
    for sel in (192, 256, -1):
        ops.set_env("POLUS_GEMM_PP", sel)
        o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm(a, b, o)
        torch.cuda.synchronize()
        outs[sel] = o.float()
    ops.set_env("POLUS_GEMM_PP")
    for sel in (192, 256):
        d = (outs[sel] != outs[-1])
        err = (outs[sel] - ref).abs().max().item()
        print(f"{M}x{N}x{K} pp{sel}: mismatches vs ring {int(d.sum())}  max|err vs f32 ref| {err:.4f} (ring {(outs[-1]-ref).abs().max().item():.4f})")
        if d.any():
            idx = d.nonzero()
            rows = idx[:, 0].unique().tolist(); cols = idx[:, 1].unique().tolist()
            print("   rows:", rows[:40], "n", len(rows)); print("   cols:", cols[:64], "n", len(cols))
            r0, c0 = idx[0].tolist()
            print("   first", r0, c0, outs[sel][r0, c0].item(), outs[-1][r0, c0].item(), ref[r0, c0].item())
