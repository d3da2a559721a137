"""Runs the attention forward and one-pass backward of one BERT-base layer (B=64, S=256, A=12, dropout 0.1) ten times each:
a target for `rocprofv3 --pmc ...` (counters only, separate from any trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops
B, S, A, H = 64, 256, 12, 768
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.rand(B * S, 3 * H, device="cuda", generator=g) * 2 - 1).bfloat16()
dctx = ((torch.rand(B * S, H, device="cuda", generator=g) * 2 - 1) * 0.1).bfloat16()
mask = torch.ones(B, S, dtype=torch.int32, device="cuda"); mask[:, 200:] = 0
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * A * S, dtype=torch.float32, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(10):
    ops.attention_fwd(qkv, mask, ctx, lse, B, S, A, drop_p=0.1, seed=5)
    ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=0.1, seed=5)
torch.cuda.synchronize()
