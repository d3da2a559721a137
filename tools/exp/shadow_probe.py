"""What a kernel gets of the chip while the grouped dW launch (216 workgroups, one per CU, 40 CUs free) is running (round 4):
the LayerNorm backward (212 VGPRs, 36 KiB LDS, 512 workgroups of 4 waves), started ~30 us after it on a second stream.
Prints each kernel alone, then the pair: wall time of the pair and the LayerNorm kernel's own duration inside it."""
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
dy, x, dx, dxm = rnd(T, H), rnd(T, H), torch.empty(T, H, dtype=dt, device=dev), torch.empty(T, H, dtype=dt, device=dev)
gam = torch.ones(H, device=dev); mean = torch.zeros(T, device=dev); rstd = torch.ones(T, device=dev)
dg, db, dbias = (torch.zeros(H, device=dev) for _ in range(3))
part = torch.empty(4 << 20, device=dev)     # room for every grid size tried below
side = torch.cuda.Stream()
DELAY = int(os.environ.get("PROBE_DELAY_CYCLES", "60000"))

def dw():
    ops.dense_bwd_params_grouped(probs, False, 0)
def ln():
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dg, db, dbias, False, dx_masked=dxm, drop_p=0.1, seed=3, partials=part)
inner = [None, None]
def both():
    ev = torch.cuda.Event(); ev.record()
    dw()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.cuda._sleep(DELAY)
        inner[0] = torch.cuda.Event(enable_timing=True); inner[1] = torch.cuda.Event(enable_timing=True)
        inner[0].record(); ln(); inner[1].record()
    torch.cuda.current_stream().wait_stream(side)

def timed(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, env in (("even split, 216 workgroups", {}), ("stream-K plan for 208 CUs", {"POLUS_DW_STREAMK": 1, "POLUS_DW_SK_CUS": 208}),
                  ("ln_bwd grid 256", {"POLUS_LN_BWD_BLOCKS": 256}), ("ln_bwd grid 1024", {"POLUS_LN_BWD_BLOCKS": 1024}),
                  ("wave-per-row ln_bwd", {"POLUS_LN_HALFWAVE": 0}), ("wave-per-row, grid 1024", {"POLUS_LN_HALFWAVE": 0, "POLUS_LN_BWD_BLOCKS": 1024}),
                  ("even split again", {})):
    for k, val in env.items():
        ops.set_env(k, val)
    a, b, c = (min(timed(f) for _ in range(3)) for f in (dw, ln, both))
    both(); torch.cuda.synchronize()
    d = inner[0].elapsed_time(inner[1]) * 1e3
    for k in env:
        ops.set_env(k)
    print(f"{name:30s} dW {a:7.1f} us   LN backward {b:6.1f} us   pair {c:7.1f} us   LN inside the pair {d:7.1f} us", flush=True)
