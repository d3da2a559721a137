"""bf16 Dense GEMM with the residual epilogue at mid-size token counts: time per launch for 1..6 K slices
(split_k > 1 = f32 slabs + splitk_reduce_epi_kernel), and what polus_gemm_auto_split recommends.  HIP-graph replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops, _lib
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)

def graph_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best

if __name__ == "__main__":
    splits = (1, 2, 3, 4, 6)
    print(f"{'M':>6s} {'N':>5s} {'K':>5s} {'auto':>5s}  " + "  ".join(f"{'s=' + str(s):>9s}" for s in splits))
    for M, N, K in ((4096, 768, 3072), (4096, 768, 2304), (2048, 768, 3072), (8192, 768, 3072), (6144, 768, 3072),
                    (4096, 1024, 4096), (4096, 1024, 1024), (8192, 1024, 4096), (2048, 1024, 4096)):
        a, b, r = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N)
        bias = torch.zeros(N, device=dev)
        c = torch.empty(M, N, dtype=dt, device=dev)
        cells = []
        for s in splits:
            t = graph_time(lambda: ops.gemm(a, b, c, bias=bias, resid=r, split_k=s))
            cells.append(f"{t*1e6:7.1f}us")
        auto = _lib.load().polus_gemm_auto_split(M, N, K)
        print(f"{M:6d} {N:5d} {K:5d} {auto:5d}  " + "  ".join(f"{x:>9s}" for x in cells), flush=True)
