"""GPU: data parallelism end to end on the real kernels.  Two ranks share the box's one GPU
(gloo wire, see dp_gpu_worker.py); after 3 AdamW steps on disjoint halves of each batch their
parameters must equal single-process training on the whole batch at LR x 2
(polus/training.py:90-94 multiplies the LR by the world size; gradients are averaged) -- for both exchange
schemes (reduce-scatter -> sharded AdamW -> all-gather, and all-reduce), with gradient accumulation, and
with bf16 gradient transport, over two epochs (the per-epoch re-broadcast of weights and optimizer variables), and
-- allreduce_epochs2 -- with a ValidationDataCallback whose per-rank GPU predictions travel through
hvd.allgather_object to rank 0's metric (polus/callbacks.py:218-261).  test_native_rccl_plane_world_1 drives every polus_comm_* entry point on a real
RCCL communicator (world size 1: the box has one GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_distributed_cpu import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["allreduce", "allreduce_split", "allreduce_accum2", "allreduce_epochs2", "allreduce_bf16", "rs", "rs_accum2", "rs_bf16", "rs_epochs2"])
def test_two_ranks_equal_one_rank_with_the_whole_batch(tmp_path, mode):
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    from tests.test_model_gpu import build_model, load_case

    out = str(tmp_path / "dp")
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, POLUS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} OK" in o, f"rank {rank} failed:\n{o}"
    r0, r1 = np.load(out + ".rank0.npy"), np.load(out + ".rank1.npy")
    assert np.array_equal(r0, r1), "ranks diverged"

    g, ocfg, params, head_w, head_b = load_case("bert_small_b2_s16")
    model = build_model(ocfg, params, head_w, head_b, "f32")
    start = model.arena.params.detach().float().cpu().numpy().copy()
    steps = 3
    epochs = 2 if mode.endswith("_epochs2") else 1
    opt = AdamWeightDecay(learning_rate=warmup_scheduler(steps * epochs, 2e-3), weight_decay_rate=0.01)
    trainer = ClassifierTrainer(model, opt, SparseCategoricalCrossentropy())
    batches = []
    for s in range(steps):
        ids, mask, tt, labels = synth_batch(ocfg, 4, 16, 4, 420 + s)
        batches.append(({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))
    trainer.train(batches, epochs=epochs, callbacks=[])
    single = model.arena.params.detach().float().cpu().numpy()
    moved = np.abs(single - start).max()
    n = min(single.size, r0.size)              # the 2-rank arena is padded to whole 64 x world element units
    assert not r0[n:].any() and not single[n:].any()
    diff = np.abs(single[:n] - r0[:n])
    # f32 sums in a different order (two half-batch gradients added on the wire vs one batch-4 GEMM);
    # Adam normalises by sqrt(v), so elements whose gradient is a near-cancelling sum carry the
    # largest relative noise: bound the worst element at 2 % of the largest move and the mean far below.
    assert moved > 1e-3
    if mode.endswith("_bf16"):      # gradients rounded to bf16 on the wire: Adam's sign-like first steps amplify it
        assert diff.mean() < 0.05 * moved, (diff.max(), diff.mean(), moved)
    else:
        assert diff.max() < 0.02 * moved and diff.mean() < 1e-3 * moved, (diff.max(), diff.mean(), moved)
    if mode == "rs_epochs2":
        # after trainer.sync_optimizer_state() both ranks hold the SAME, whole Adam moments, and they are the
        # single-process moments (the per-epoch broadcast from rank 0 must not have replaced rank 1's slices
        # with rank 0's never-updated copies of them)
        sm, sv = (t.detach().float().cpu().numpy() for t in opt._slots(model.arena))
        for name, ref in (("m", sm), ("v", sv)):
            a, b = np.load(f"{out}.rank0.{name}.npy"), np.load(f"{out}.rank1.{name}.npy")
            assert np.array_equal(a, b), f"Adam {name} differs between the ranks after the sync"
            assert np.abs(a[:n]).max() > 0 and not a[n:].any()
            err = np.abs(a[:n] - ref[:n])
            assert err.max() < 0.02 * np.abs(ref[:n]).max() and err.mean() < 2e-3 * np.abs(ref[:n]).mean(), (name, err.max(), err.mean())
        assert os.path.exists(out + ".rank0.state.npz")


def test_native_rccl_plane_world_1():
    """The C-ABI collectives (include/polus_hip.h polus_comm_*) on a real RCCL communicator of size 1: id,
    init, all-reduce, reduce-scatter, all-gather (in place, as the reducer issues them), broadcast, destroy --
    queued on a side stream and fenced with events exactly as comm._NativePlane does for N ranks."""
    import ctypes
    import torch
    from polus_amd import _lib
    from polus_amd.comm import _NativePlane
    lib = _lib.load()
    uid = (ctypes.c_ubyte * 128)()
    _lib.check(lib.polus_comm_unique_id(uid), "polus_comm_unique_id")
    assert any(uid), "unique id is all zeros"
    plane = _NativePlane.__new__(_NativePlane)
    plane.lib, plane.check = lib, _lib.check
    plane.comm = ctypes.c_void_p()
    _lib.check(lib.polus_comm_init(ctypes.byref(plane.comm), 0, 1, uid), "polus_comm_init")
    plane.stream, plane.world, plane.rank = torch.cuda.Stream(), 1, 0
    x = torch.arange(1 << 16, dtype=torch.float32, device="cuda")
    ref = x.clone()
    plane.all_reduce_sum(x).wait()
    plane.reduce_scatter_sum(x, x).wait()
    plane.all_gather(x, x).wait()
    plane.broadcast(x, 0)
    h = torch.arange(4096, dtype=torch.float32, device="cuda").to(torch.bfloat16)
    plane.all_reduce_sum(h).wait()
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(h.float(), torch.arange(4096, dtype=torch.float32, device="cuda").to(torch.bfloat16).float())
    assert lib.polus_comm_broadcast(plane.comm, x.data_ptr(), 16, 3, None) != 0 and b"bad root" in lib.polus_last_error()
    plane.close()


def test_bench_bare_multi_gpu_launch_line(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it: the parent spawns both ranks (before it touches torch or
    the GPU), rank 0 prints the one JSON line, and the N > 1 line describes its own exchange: how many ranks the data
    plane really spanned, which plane, the exposed tail behind backward and the step time with no exchange at all.
    Two ranks share the box's one GPU, so the wire is gloo (RCCL refuses duplicate devices)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(POLUS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--layers", "2", "--batch", "8", "--seq", "64", "--no-f32-leg", "--no-loss100", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["data_plane"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["exposed_comm_ms"] > 0 and out["ms_per_step_nocomm"] > 0 and out["ms_per_step"] > 0
    assert out["value"] > 0 and out["roofline"]["launches"] > 0 and "unpinned" in out["multi_gpu_parity"]
    # the measured choice of the CU reserve ran on both ranks and both report the same decision (max-over-ranks times)
    t = out["dp_tuning"]
    assert t["reserve_on_ms"] > 0 and t["reserve_off_ms"] > 0
    assert t["reserve_cus_in_backward"] == (t["reserve_on_ms"] <= t["reserve_off_ms"])
    assert "NCCL_MAX_NCHANNELS" in t and "POLUS_GEMM_RESERVE_CUS" in t       # the settings the exchange ran with, as used
    assert out["cpu_baseline"]["value"] is None and "N=1" in out["cpu_baseline"]["note"]
    sizes = {mb: t[f"bucket_{mb}_ms"] for mb in (16, 32, 64, 128)}
    assert all(v > 0 for v in sizes.values()) and t["bucket_mb"] in sizes and sizes[t["bucket_mb"]] == min(sizes.values())
