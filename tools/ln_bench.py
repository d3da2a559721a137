"""LayerNorm forward / backward at the headline shape: wave-per-row (8-byte accesses) vs half-wave-per-row (16-byte)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from tools.gemm_bench import bench
T, H = 16384, 768
dev, dt = "cuda", torch.bfloat16
x = torch.randn(T, H, device=dev).to(dt); dy = torch.randn(T, H, device=dev).to(dt)
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
y, dx, dxm = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
dg, db, dbias = torch.empty(H, device=dev), torch.empty(H, device=dev), torch.empty(H, device=dev)
for hw, blocks in ((0, 1024), (1, 1024), (1, 512), (1, 256)):
    ops.set_env("POLUS_LN_HALFWAVE", hw); ops.set_env("POLUS_LN_BWD_BLOCKS", blocks)
    tf = bench(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-12), 30)
    tb = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, dbias, dx_masked=dxm, drop_p=0.1, seed=3), 30)
    print(f"halfwave={hw} bwd blocks<={blocks}: fwd {tf*1e6:.1f} us ({2*T*H*2/tf/1e12:.2f} TB/s)   bwd+finalize {tb*1e6:.1f} us ({4*T*H*2/tb/1e12:.2f} TB/s)", flush=True)
ops.set_env("POLUS_LN_HALFWAVE"); ops.set_env("POLUS_LN_BWD_BLOCKS")
