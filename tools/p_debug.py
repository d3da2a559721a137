import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
os.environ["POLUS_GEMM_P"] = "2"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)
for (M, N, K) in [(256, 192, 768), (256, 192, 1536), (256, 192, 1600), (256, 192, 3072), (512, 192, 3072), (256, 384, 3072), (1024, 192, 3072), (256*260, 192, 128)]:
    a = rnd(M, K).bfloat16(); b = (rnd(N, K) * 0.1).bfloat16()
    ref = a.float() @ b.float().t()
    c = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, b, c); torch.cuda.synchronize()
    d = (c.float() - ref).abs()
    bad = d > 0.1
    print(M, N, K, "max err", d.max().item(), "bad frac", bad.float().mean().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   bad rows", rows[:8].tolist(), "...", rows[-3:].tolist(), len(rows), " bad cols", cols[:8].tolist(), len(cols))
        # which k-range is missing? compare against partial sums
        for k1 in range(32, K + 1, 32):
            part = a[:, :k1].float() @ b[:, :k1].float().t()
            if (c.float() - part).abs()[bad].max() < 0.1:
                print("   matches partial sum up to k =", k1); break
