"""How long the HOST takes to queue a training step against how long the GPU takes to run it (headline shape): if the two are close,
the gaps at step boundaries in a kernel trace are the host's.  python tools/host_time.py [--graph]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
args = A()
args.kind, args.accum, args.batch, args.seq, args.geom, args.dropout, args.graph = "ner", 1, 64, 256, (768, 12, 3072, 12), 0.1, "--graph" in sys.argv
args.config = "c3"
torch.cuda.set_device(0)
model, trainer = bench.build_trainer(args, "bf16", 0.1, 1000)
if args.graph:
    trainer.enable_step_graph(warmup=2)
batches = bench.device_batches(args, 0, model.arena.device)
step = bench.make_step(args, trainer, batches)
for k in range(6):
    step(k)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for k in range(N):
    step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph={args.graph}: host queued {N} steps in {(t1 - t0) / N * 1e3:.3f} ms/step; GPU done after {(t2 - t0) / N * 1e3:.3f} ms/step", flush=True)
