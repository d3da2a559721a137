"""Half-wave vs wave-per-row LayerNorm on the same bf16 input, each against an f64 reference (numpy), at the
BERT-large shapes of configs[3]; then the c4 bf16 loss under either kernel.  Prints only; no assertions."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops

def ref64(x, g, b, eps):
    x = x.astype(np.float64); m = x.mean(1, keepdims=True); v = ((x - m) ** 2).mean(1, keepdims=True)
    return (x - m) / np.sqrt(v + eps) * g + b, m[:, 0], 1 / np.sqrt(v[:, 0] + eps)

for T, H, scale in ((4096, 1024, 1.0), (4096, 1024, 8.0), (512, 1024, 1.0), (4096, 768, 1.0), (4098, 1024, 1.0)):
    torch.manual_seed(T + H)
    x = (torch.randn(T, H, device="cuda") * scale + 0.3).to(torch.bfloat16)
    g = torch.randn(H, device="cuda") * 0.2 + 1; b = torch.randn(H, device="cuda") * 0.1
    r, rm, rr = ref64(x.float().cpu().numpy(), g.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64), 1e-12)
    out = {}
    for hw in (0, 1):
        ops.set_env("POLUS_LN_HALFWAVE", hw)
        y = torch.full_like(x, float("nan")); mean = torch.full((T,), float("nan"), device="cuda"); rstd = mean.clone()
        ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-12); torch.cuda.synchronize()
        yy = y.float().cpu().numpy().astype(np.float64)
        out[hw] = y
        print(f"T={T} H={H} scale={scale} hw={hw}: max|y-ref| {np.abs(yy - r).max():.4e}  rms {np.sqrt(((yy - r) ** 2).mean()):.4e}  "
              f"max|mean-ref| {np.abs(mean.cpu().numpy() - rm).max():.3e}  max rel rstd {np.abs(rstd.cpu().numpy() / rr - 1).max():.3e}  "
              f"nan {int(np.isnan(yy).sum())}", flush=True)
    d = (out[0].float() - out[1].float()).abs()
    rows = (d.max(1).values > 0).sum().item()
    print(f"    hw0 vs hw1: {int((d > 0).sum())} of {d.numel()} elements differ, in {rows} rows; max diff {d.max().item():.4e}", flush=True)
ops.set_env("POLUS_LN_HALFWAVE")
