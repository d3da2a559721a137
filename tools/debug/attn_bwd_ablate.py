"""Key-resident attention backward with parts switched off (POLUS_ATTN_DEBUG bits: 1 no dQ phase, 2 no element-wise part,
4 no dK/dV MFMAs, 8 no delta loop, 16 no dQ row stores): where the time of a slice goes.  Results are wrong by design."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops
from tools.gemm_bench import bench
B, S, A = 64, 256, 12
H = A * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = ((torch.rand(B * S, 3 * H, device="cuda", generator=g) * 2 - 1)).bfloat16()
dctx = ((torch.rand(B * S, H, device="cuda", generator=g) * 2 - 1) * 0.1).bfloat16()
mask = torch.ones(B, S, dtype=torch.int32, device="cuda"); mask[:, 200:] = 0
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * A * S, dtype=torch.float32, device="cuda")
dqkv = torch.empty_like(qkv)
ops.attention_fwd(qkv, mask, ctx, lse, B, S, A)
ops.set_env("POLUS_ATTN_BWD_KRES", 2)
for p in (0.0, 0.1):
    for dbg in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31):
        ops.set_env("POLUS_ATTN_DEBUG", dbg)
        t = bench(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, A, drop_p=p, seed=5), 20)
        print(f"drop {p} debug {dbg:2d}: {t*1e6:6.1f} us", flush=True)
ops.set_env("POLUS_ATTN_DEBUG")
ops.set_env("POLUS_ATTN_BWD_KRES")
