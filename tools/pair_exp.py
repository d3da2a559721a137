import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from tools.gemm_bench import bench
dt = torch.bfloat16
for (M, N, K) in [(8192, 2048, 768), (8192, 2048, 3072), (16384, 1024, 1024)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt); b = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(dt)
    c = torch.empty(M, N, dtype=dt, device="cuda")
    fl = 2.0 * M * N * K
    for ab in ("0", "4"):
        os.environ["POLUS_GEMM_ABLATE"] = ab
        t = bench(lambda: ops.gemm(a, b, c), 20)
        print(M, N, K, "pair" if ab == "4" else "base", f"{fl / t / 1e12:.1f} TF/s {t * 1e6:.1f} us")
del os.environ["POLUS_GEMM_ABLATE"]
