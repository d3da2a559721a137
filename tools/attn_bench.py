"""Attention kernels of one BERT-base layer (B=64, S=256, A=12) with and without dropout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from tools.gemm_bench import bench
B, S, A, H = 64, 256, 12, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = ((torch.rand(B * S, 3 * H, device=dev, generator=g) * 2 - 1)).bfloat16()
dctx = ((torch.rand(B * S, H, device=dev, generator=g) * 2 - 1) * 0.1).bfloat16()
mask = torch.ones(B, S, dtype=torch.int32, device=dev); mask[:, 200:] = 0
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * A * S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv)
for p in (0.0, 0.1):
    tf = bench(lambda: ops.attention_fwd(qkv, mask, ctx, lse, B, S, A, drop_p=p, seed=5), 20)
    tb = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
    ops.set_env("POLUS_ATTN_FUSED", 0)
    t2 = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
    ops.set_env("POLUS_ATTN_FUSED")
    print(f"drop_p={p}: fwd {tf*1e6:.1f} us   bwd one pass {tb*1e6:.1f} us   bwd two kernels (dq + dkv) {t2*1e6:.1f} us", flush=True)
