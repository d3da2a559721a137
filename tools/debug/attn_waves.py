import os, sys, torch
sys.path.insert(0, '/root/repo')
from polus_amd import ops
from tools.gemm_bench import bench
B, S, A, H = 64, 256, 12, 768
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.rand(B * S, 3 * H, device="cuda", generator=g) * 2 - 1).bfloat16()
mask = torch.ones(B, S, dtype=torch.int32, device="cuda"); mask[:, 200:] = 0
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * A * S, dtype=torch.float32, device="cuda")
for rep in range(2):
    for w in (4, 8, 16):
        ops.set_env("POLUS_ATTN_WAVES", w)
        t = bench(lambda: ops.attention_fwd(qkv, mask, ctx, lse, B, S, A, drop_p=0.1, seed=5), 30)
        print(f"POLUS_ATTN_WAVES={w}: fwd {t*1e6:.1f} us", flush=True)
ops.set_env("POLUS_ATTN_WAVES")
