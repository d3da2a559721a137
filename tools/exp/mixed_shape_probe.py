"""Probe: FFN1 / dU as TWO concurrent persistent launches on two streams -- rows 0..M/2 on 128 CUs with 256-wide tiles
(3 per CU), rows M/2..M on 128 CUs with 192-wide tiles (4 per CU) -- so that the epilogue bursts of the two halves
interleave instead of hitting HBM together.  Compared with the single persistent launch, cold caches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops, _lib
T, dt = 16384, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device="cuda", generator=g) * 2 - 1).to(dt)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
for name, N, K, mode in (("ffn1", 3072, 768, "gelu"), ("du", 3072, 768, "bwd")):
    a, b = rnd(T, K), rnd(N, K) * 0.05
    c = torch.empty(T, N, dtype=dt, device="cuda")
    bias = torch.zeros(N, device="cuda")
    aux = rnd(T, N)
    kw = dict(bias=bias, aux=aux, act="gelu", flags=ops.GEMM_ACT_FWD) if mode == "gelu" else dict(aux=aux, act="gelu", flags=ops.GEMM_ACT_BWD)
    h = T // 2

    def single():
        ops.gemm(a, b, c, **kw)

    def mixed():
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        kw1 = dict(kw); kw1["aux"] = aux[:h]
        kw2 = dict(kw); kw2["aux"] = aux[h:]
        ops.set_env("POLUS_GEMM_PP", 256)
        ops.gemm(a[:h], b, c[:h], **kw1)
        ops.set_env("POLUS_GEMM_PP", 192)
        with _lib.stream_scope(side):
            ops.gemm(a[h:], b, c[h:], **kw2)
        ev2 = torch.cuda.Event(); ev2.record(side)
        torch.cuda.current_stream().wait_event(ev2)

    def graphed(fn):
        """The two launches replayed from a captured HIP graph: no host launch latency between them."""
        fn(); torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            with _lib.pinned_stream():
                fn()
        return g_.replay

    def timed(fn, iters=8):
        tot = 0.0
        for _ in range(iters):
            scrub.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / iters * 1e3
    ops.set_env("POLUS_GEMM_PERSIST", 1); ops.set_env("POLUS_GEMM_RESERVE_CUS"); ops.set_env("POLUS_GEMM_PP")
    single(); t_single = [timed(single) for _ in range(3)]
    t_single_g = [timed(graphed(single)) for _ in range(3)]
    ops.set_env("POLUS_GEMM_PERSIST", 2); ops.set_env("POLUS_GEMM_RESERVE_CUS", 128)
    mixed(); t_mixed = [timed(mixed) for _ in range(3)]
    t_mixed_g = [timed(graphed(mixed)) for _ in range(3)]
    ops.set_env("POLUS_GEMM_PP", 256)
    def same():   # both halves 256-wide: concurrency alone, no shape mix
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        kw1 = dict(kw); kw1["aux"] = aux[:h]
        kw2 = dict(kw); kw2["aux"] = aux[h:]
        ops.gemm(a[:h], b, c[:h], **kw1)
        with _lib.stream_scope(side):
            ops.gemm(a[h:], b, c[h:], **kw2)
        ev2 = torch.cuda.Event(); ev2.record(side); torch.cuda.current_stream().wait_event(ev2)
    same(); t_same = [timed(same) for _ in range(3)]
    t_same_g = [timed(graphed(same)) for _ in range(3)]
    ops.set_env("POLUS_GEMM_PERSIST"); ops.set_env("POLUS_GEMM_RESERVE_CUS"); ops.set_env("POLUS_GEMM_PP")
    print(f"{name}: single persistent launch {min(t_single):.1f} us (graph replay {min(t_single_g):.1f})   two halves 256 | 192 on two streams {min(t_mixed):.1f} us "
          f"(graph replay {min(t_mixed_g):.1f})   two halves 256 | 256 {min(t_same):.1f} us (graph replay {min(t_same_g):.1f})", flush=True)
