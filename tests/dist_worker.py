"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job on CPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bert as ob  # noqa: E402
from oracle import optim as oo  # noqa: E402
from polus_amd import comm  # noqa: E402
from polus_amd.context import PolusContext  # noqa: E402
from polus_amd.data import shard  # noqa: E402
from polus_amd.training import ClassifierTrainer  # noqa: E402
from tests.fakes import FakeLinearModel, FakeSGD, FakeXent  # noqa: E402


def main():
    ctx = PolusContext()
    assert ctx.is_horovod_enabled() and comm.size() == 2 and ctx.backend == "gloo"
    rank = comm.rank()
    assert comm.local_rank() == rank

    # -- broadcast + allgather_object (C2/C3/C4 of SURVEY.md §2c)
    m = FakeLinearModel(seed=rank)
    comm.broadcast_variables(m.trainable_weights, root_rank=0)
    ref = FakeLinearModel(seed=0)
    assert torch.equal(m.arena.params, ref.arena.params)
    got = comm.allgather_object((np.array([rank, rank + 10]), f"r{rank}"))
    assert [g[1] for g in got] == ["r0", "r1"] and got[1][0].tolist() == [1, 11]

    # -- bucketed reducer: SUM over ranks, buckets launched in descending order as they get ready
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 300, boundaries=[0, 100, 250, 600, 900])
    r.begin()
    r.on_ready(900, 1000)
    assert r._next == 1
    r.on_ready(250, 900)
    assert r._next == 3
    r.finish()
    assert r._next == 4 and torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3)

    # -- the same arena through reduce-scatter -> update of the owned slices -> all-gather (the "rs" scheme):
    #    equal buckets of whole 64 x world element units from the end, rank r owns slice r of every bucket
    n = 64 * 2 * 9
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 64 * 2 * 2, mode="rs")
    assert r.buckets[0] == (n - 256, n) and r.buckets[-1][0] == 0 and all((hi - lo) % 128 == 0 for lo, hi in r.buckets)
    r.begin()
    r.on_ready(n - 256, n)
    assert r._next == 1
    r.on_ready(n - 300, n - 256)                 # not a whole bucket yet
    assert r._next == 1
    r.finish()
    owned = r.owned_ranges()
    assert sum(hi - lo for lo, hi in owned) == n // 2
    full = torch.arange(n, dtype=torch.float32) * 3
    for lo, hi in owned:
        assert torch.equal(g[lo:hi], full[lo:hi]), (lo, hi)
    p = torch.full((n,), -1.0)
    for lo, hi in owned:                          # "optimizer step" on the owned slices only
        p[lo:hi] = -0.5 * g[lo:hi]
    r.allgather(p)
    assert torch.equal(p, -0.5 * full)

    # -- data parallel == large batch, with the oracle as the per-rank compute
    cfg = ob.BertConfig(40, 64, 1, 1, 128, 16, 2)
    params, hw, hb = ob.golden_setup(cfg, 3)
    rng = np.random.Generator(np.random.PCG64(5))
    ids = rng.integers(0, 40, size=(4, 8)).astype(np.int32)
    mask = np.ones((4, 8), np.int32); mask[1, 6:] = 0
    labels = rng.integers(0, 3, size=(4, 8)).astype(np.int32)
    mine = list(shard(range(4), 2, comm.local_rank()))          # sample i -> rank i mod 2
    assert mine == oo.shard_indices(4, 2, rank)
    _, _, cache = ob.token_classifier_fwd(params, cfg, hw, hb, ids[mine], mask[mine], labels[mine])
    gl = ob.token_classifier_bwd(params, cfg, hw, cache)
    names = sorted(gl)
    flat = torch.from_numpy(np.concatenate([gl[k].reshape(-1) for k in names]).astype(np.float64))
    offs = np.cumsum([0] + [gl[k].size for k in names])[:-1].tolist()
    red = comm.GradBucketReducer(flat, bucket_bytes=8 * 4096, boundaries=offs)
    red.begin()
    red.finish()
    flat *= 1.0 / comm.size()
    _, _, cache = ob.token_classifier_fwd(params, cfg, hw, hb, ids, mask, labels)
    gf = ob.token_classifier_bwd(params, cfg, hw, cache)
    full = np.concatenate([gf[k].reshape(-1) for k in names])
    assert np.abs(flat.numpy() - full).max() < 1e-12

    # -- trainer in DP mode: LR x world (polus/training.py:90-94), gradients averaged,
    #    weights broadcast from rank 0 at step 0 of the epoch (:318-319)
    model = FakeLinearModel(seed=rank)
    trainer = ClassifierTrainer(model, FakeSGD(0.1), FakeXent())
    assert trainer.use_horovod and abs(trainer.optimizer.learning_rate(0) - 0.2) < 1e-12
    r2 = np.random.default_rng(100 + rank)
    x, y = r2.standard_normal((4, 5)), r2.integers(0, 3, size=4)
    trainer.train([(x, y)], epochs=1)
    # expected: w0(rank 0) - 2*lr * mean over ranks of local gradients
    w0 = FakeLinearModel(seed=0)
    gs = []
    for rk in range(2):
        rr = np.random.default_rng(100 + rk)
        xx, yy = rr.standard_normal((4, 5)), rr.integers(0, 3, size=4)
        mm = FakeLinearModel(seed=0)
        f = FakeXent(); f(yy, mm(xx)); mm.backward(f.backward())
        gs.append(mm.arena.grads.clone())
    expect = w0.arena.params - 0.2 * (gs[0] + gs[1]) / 2
    assert torch.allclose(model.arena.params, expect, atol=1e-6), (model.arena.params, expect)
    comm.barrier()
    comm.shutdown()
    print(f"rank {rank} OK")


if __name__ == "__main__":
    main()
