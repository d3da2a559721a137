"""dX = dY . W: W[out, in] as the K-strided B operand (what the model does) against a transposed
copy W^T[in, out] as a K-contiguous B operand, on the four dX shapes of a BERT-base layer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd import ops
from tools.gemm_bench import bench
T = 16384
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).bfloat16()
# (name, out_features = K of the product, in_features = N of the product, epilogue)
for name, n_out, n_in, epi in [("dqkv->dx", 2304, 768, "resid"), ("dz1->dctx", 768, 768, "plain"), ("dz2->du", 768, 3072, "gelu-bwd"), ("du->da1", 3072, 768, "resid")]:
    dy = rnd(T, n_out); w = rnd(n_out, n_in) * 0.05; wt = w.t().contiguous()
    dx = torch.empty(T, n_in, dtype=torch.bfloat16, device=dev)
    res = rnd(T, n_in); u = rnd(T, n_in)
    kw = {"resid": dict(resid=res), "plain": {}, "gelu-bwd": dict(aux=u, act="gelu", flags=ops.GEMM_ACT_BWD)}[epi]
    fl = 2.0 * T * n_out * n_in
    t_ks = bench(lambda: ops.gemm(dy, w, dx, b_layout=ops.K_STRIDED, **kw), 20)
    t_kc = bench(lambda: ops.gemm(dy, wt, dx, **kw), 20)
    t_tr = bench(lambda: ops.transpose_bf16(w, wt), 20)
    print(f"{name:10s} K={n_out:5d} N={n_in:5d} {epi:9s}  K-strided {t_ks*1e6:7.1f} us {fl/t_ks/1e12:6.1f} TF   K-contig(W^T) {t_kc*1e6:7.1f} us {fl/t_kc/1e12:6.1f} TF   transpose {t_tr*1e6:5.1f} us", flush=True)
