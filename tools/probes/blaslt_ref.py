"""Reference point only (not used by the product): what the vendor GEMM reaches on the
forward Dense shapes, and which macro-tile it picks (kernel names via rocprofv3)."""
import torch
T = 16384
shapes = [("qkv", T, 2304, 768), ("out", T, 768, 768), ("ffn1", T, 3072, 768), ("ffn2", T, 768, 3072)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(a, b.t(), out=c)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{name} {M}x{N}x{K}: {2*M*N*K/t/1e12:.1f} TF/s {t*1e6:.1f} us", flush=True)
