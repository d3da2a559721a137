"""Attention kernels of one BERT-base layer (B=64, S=256, A=12) with and without dropout; forward A/B of the LDS-DMA
kernel against the register-staged one (POLUS_ATTN_FWD_DMA=0), interleaved rounds in one process.

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
    t = {1: [], 0: []}
    outs = {}
    for rnd in range(5):
        for v in (1, 0):
            ops.set_env("POLUS_ATTN_FWD_DMA", v)
            t[v].append(bench(fwd, 20))
            if rnd == 0:
                outs[v] = (ctx.float().clone(), lse.clone())
    ops.set_env("POLUS_ATTN_FWD_DMA")
    dc = (outs[1][0] - outs[0][0]).abs().max().item()
    dl = (outs[1][1] - outs[0][1]).abs().max().item()
    ops.set_env("POLUS_ATTN_BWD_KRES", 2)
    tb = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
    ops.set_env("POLUS_ATTN_BWD_KRES", 0)
    ops.set_env("POLUS_ATTN_FUSED", 3)
    tq = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
    ops.set_env("POLUS_ATTN_FUSED", 2)
    t64 = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20) if S in (64, 128, 256) else float("nan")
    ops.set_env("POLUS_ATTN_FUSED")
    ops.set_env("POLUS_ATTN_BWD_KRES")
    t2 = None
    if S in (64, 128, 256):
        ops.set_env("POLUS_ATTN_FUSED", 0)
        t2 = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
        ops.set_env("POLUS_ATTN_FUSED")
    med = lambda x: sorted(x)[len(x) // 2] * 1e6
    print(f"B={B} S={S} drop_p={p}: fwd LDS-DMA {med(t[1]):.1f} us (min {min(t[1])*1e6:.1f})  register-staged {med(t[0]):.1f} us   "
          f"|ctx diff| {dc:.2e} |lse diff| {dl:.2e}   bwd key-resident {tb*1e6:.1f} us (query-resident, 32-key blocks: {tq*1e6:.1f} us; 64-key blocks: {t64*1e6:.1f} us)" + (f"   bwd two kernels {t2*1e6:.1f} us" if t2 else ""), flush=True)
