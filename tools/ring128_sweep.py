"""The 128 x 128 ring tile against the other paths of a bf16 Dense GEMM at mid-size token counts: correctness of every
epilogue mode against the 256 x 128 ring kernel (bit-identical: same k order, same epilogue arithmetic) and time per launch
(HIP-graph replay) of: default dispatch with no slicing, forced 128 x 128, and the best K slicing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from polus_amd._lib import GEMM_ACT_FWD, GEMM_ACT_BWD
from tools.split_sweep import graph_time
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)

def check(M, N, K):
    a, b, r, u = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N), rnd(M, N)
    bias = torch.rand(N, device=dev, generator=g)
    cases = {"bias": dict(bias=bias), "gelu+aux": dict(bias=bias, act="gelu", flags=GEMM_ACT_FWD, aux="new"),
             "resid": dict(bias=bias, resid=r), "gelu'": dict(aux=u, act="gelu", flags=GEMM_ACT_BWD),
             "drop+resid": dict(bias=bias, resid=r, drop_p=0.2, seed=11)}
    ok = True
    for name, kw in cases.items():
        outs = []
        for sel in (-1, 1):
            ops.set_env("POLUS_GEMM_RING128", sel); ops.set_env("POLUS_GEMM_PP", -1)
            c = torch.full((M, N), float("nan"), dtype=dt, device=dev)
            kw2 = dict(kw)
            if kw2.get("aux") == "new":
                kw2["aux"] = torch.full((M, N), float("nan"), dtype=dt, device=dev)
            ops.gemm(a, b, c, split_k=1, **kw2); torch.cuda.synchronize()
            outs.append((c, kw2.get("aux") if kw.get("aux") == "new" else None))
        same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
        ok &= same
        if not same:
            d = (outs[0][0].float() - outs[1][0].float()).abs()
            print(f"   MISMATCH {name} at {M}x{N}x{K}: {int((d > 0).sum())} elements, max {d.max().item():.3e}, nan {int(torch.isnan(outs[1][0]).sum())}")
    ops.set_env("POLUS_GEMM_RING128"); ops.set_env("POLUS_GEMM_PP")
    return ok

if __name__ == "__main__":
    for shape in ((512, 768, 768), (1000, 1000, 1056), (4096, 768, 2304), (384, 256, 96)):
        print(f"correctness {shape}: {'bit-identical to the 256-row ring tile in all 5 modes' if check(*shape) else 'FAILED'}", flush=True)
    print(f"{'M':>6s} {'N':>5s} {'K':>5s}  {'default':>9s} {'ring128':>9s}   best slicing")
    for M, N, K in ((4096, 768, 3072), (4096, 768, 2304), (4096, 768, 768), (8192, 768, 3072), (8192, 768, 768), (6144, 768, 3072),
                    (4096, 1024, 4096), (4096, 1024, 1024), (8192, 1024, 4096), (2048, 768, 3072), (16384, 768, 768)):
        a, b, r = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N)
        bias = torch.zeros(N, device=dev)
        c = torch.empty(M, N, dtype=dt, device=dev)
        ops.set_env("POLUS_GEMM_RING128", -1)
        t_def = graph_time(lambda: ops.gemm(a, b, c, bias=bias, resid=r, split_k=1))
        best = min(((graph_time(lambda: ops.gemm(a, b, c, bias=bias, resid=r, split_k=s)), s) for s in (2, 3, 4) if K // s >= 512), default=(float("nan"), 0))
        ops.set_env("POLUS_GEMM_RING128", 1); ops.set_env("POLUS_GEMM_PP", -1)
        t_128 = graph_time(lambda: ops.gemm(a, b, c, bias=bias, resid=r, split_k=1))
        ops.set_env("POLUS_GEMM_RING128"); ops.set_env("POLUS_GEMM_PP")
        print(f"{M:6d} {N:5d} {K:5d}  {t_def*1e6:7.1f}us {t_128*1e6:7.1f}us   {best[0]*1e6:7.1f}us at {best[1]} slices", flush=True)
