"""Attention kernels of one BERT-base layer (B=64, S=256, A=12) with and without dropout; forward A/B of the LDS-DMA
kernel, and the backward forms of the shape (the default one, and the alternatives POLUS_ATTN_BWD_KRES / POLUS_ATTN_FUSED select).

    python tools/attn_bench.py [--seq 256] [--batch 64]
"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from tools.gemm_bench import bench
ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--heads", type=int, default=12)
args = ap.parse_args()
B, S, A = args.batch, args.seq, args.heads
H = A * 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = ((torch.rand(B * S, 3 * H, device=dev, generator=g) * 2 - 1)).bfloat16()
dctx = ((torch.rand(B * S, H, device=dev, generator=g) * 2 - 1) * 0.1).bfloat16()
mask = torch.ones(B, S, dtype=torch.int32, device=dev); mask[:, int(S * 0.78):] = 0
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * A * S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv)
for p in (0.0, 0.1):
    fwd = lambda: ops.attention_fwd(qkv, mask, ctx, lse, B, S, A, drop_p=p, seed=5)
    bwd = lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5)
    tf = [bench(fwd, 20) for _ in range(5)]
    td = [bench(bwd, 20) for _ in range(5)]                    # the default backward of this shape
    extra = ""
    if S % 256 == 0 and S <= 2048:
        ops.set_env("POLUS_ATTN_BWD_KRES", 2 if S == 256 else 0)     # the other one-pass form of this shape
        extra += f"   bwd {'key-resident' if S == 256 else 'two kernels'} {bench(bwd, 20) * 1e6:.1f} us"
        ops.set_env("POLUS_ATTN_BWD_KRES")
    if S in (64, 128, 256):
        ops.set_env("POLUS_ATTN_FUSED", 0)
        extra += f"   bwd two kernels {bench(bwd, 20) * 1e6:.1f} us"
        ops.set_env("POLUS_ATTN_FUSED")
    med = lambda x: sorted(x)[len(x) // 2] * 1e6
    print(f"B={B} S={S} drop_p={p}: fwd {med(tf):.1f} us (min {min(tf) * 1e6:.1f})   bwd {med(td):.1f} us (min {min(td) * 1e6:.1f})" + extra, flush=True)
