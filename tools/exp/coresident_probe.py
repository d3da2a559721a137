"""Does a small-register memory-bound kernel share CUs with the grouped dW launch?  (round 4)
dW alone, a 213 MB AdamW window alone, both on two streams; with the even K split (216 workgroups on 256 CUs) and with
the stream-K remainder (all 256 CUs).  If the pair takes about max(a, b) even when dW holds every CU, the AdamW waves
(62 VGPRs) are resident beside the dW workgroups (2 x 224 of 512 registers per SIMD, 128 of 160 KiB LDS)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polus_amd import ops

T, H, I = 16384, 768, 3072
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.rand(r, c, device=dev, generator=g) * 2 - 1).to(dt)
shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
probs = [(rnd(T, o), rnd(T, i), torch.empty(o, i, device=dev), torch.empty(o, device=dev)) for o, i in shapes]
n = 7087872
NW = 8                                       # the AdamW window rotates over 8 layers' worth of state: every launch reads HBM, as in the step
p, gr, m, v = (torch.zeros(n * NW, device=dev) for _ in range(4))
sh = torch.zeros(n * NW, dtype=dt, device=dev)
CH = 1 << 14
segs = [torch.tensor([[w * n + b, w * n + min(n, b + CH), 3] for b in range(0, n, CH)], dtype=torch.int64, device=dev) for w in range(NW)]
NSEG = segs[0].shape[0]
turn = [0]
DELAY = int(os.environ.get("PROBE_DELAY_CYCLES", "60000"))      # ~30 us: the dW workgroups are resident before the update is dispatched
side = torch.cuda.Stream()

def dw():
    ops.dense_bwd_params_grouped(probs, False, 0)
def adam():
    turn[0] = (turn[0] + 1) % NW
    ops.adam_step(p, gr, m, v, sh, segs[turn[0]], NSEG, 1e-4, 1e-4, 0.9, 0.999, 1e-6, 0.01)
def both():
    ev = torch.cuda.Event(); ev.record()
    dw()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.cuda._sleep(DELAY)
        adam()
    torch.cuda.current_stream().wait_stream(side)

def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, env in (("even split, 216 workgroups", {}), ("stream-K remainder, 256 workgroups", {"POLUS_DW_STREAMK": 1})):
    for k, val in env.items():
        ops.set_env(k, val)
    a, b, c = (min(timed(f) for _ in range(3)) for f in (dw, adam, both))
    for k in env:
        ops.set_env(k)
    print(f"{name:38s} dW {a:7.1f} us   AdamW window {b:6.1f} us   both on two streams {c:7.1f} us   (sum {a + b:7.1f})", flush=True)
