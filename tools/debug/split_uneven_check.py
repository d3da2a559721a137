import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from polus_amd import ops
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
for M, N, K, s in ((4096, 1024, 4096, 3), (512, 768, 2304, 5), (1024, 1024, 4096, 7)):
    a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dt); b = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * 0.05).to(dt)
    r = (torch.rand(M, N, device="cuda", generator=g) * 2 - 1).to(dt); bias = torch.rand(N, device="cuda", generator=g)
    ref = a.double() @ b.double().T + bias.double() + r.double()
    outs = {}
    for sk in (1, s):
        c = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
        ops.gemm(a, b, c, bias=bias, resid=r, split_k=sk); torch.cuda.synchronize()
        outs[sk] = c
        e = (c.double() - ref).abs()
        print(f"M={M} N={N} K={K} split={sk}: max err {e.max().item():.4e}  rms {e.pow(2).mean().sqrt().item():.4e}  nan {int(torch.isnan(c).sum())}", flush=True)
    d = (outs[1].float() - outs[s].float()).abs()
    print(f"    unsplit vs split: {int((d > 0).sum())} of {d.numel()} differ, max {d.max().item():.4e}")
