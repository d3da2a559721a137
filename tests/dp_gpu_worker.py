"""Worker of tests/test_dp_gpu.py: one rank of a 2-rank data-parallel job whose ranks share
the box's single MI355X (gloo transport: RCCL refuses two ranks on one device).  Everything
but the wire -- kernels, arenas, bucket reducer, Adam's folded 1/N, LR x world, the step-0
broadcast -- is the product path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "rs"             # (rs | allreduce)[_accum2 | _epochs2 | _bf16]
    scheme = "allreduce" if mode.startswith("allreduce") else "rs"
    os.environ["POLUS_DP_MODE"] = scheme
    if mode.endswith("_bf16"):
        os.environ["POLUS_DP_BF16"] = "1"
    from polus_amd import comm
    from polus_amd.context import PolusContext
    from polus_amd.data import shard
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    from tests.test_model_gpu import build_model, load_case

    ctx = PolusContext()
    assert ctx.is_horovod_enabled() and comm.size() == 2
    rank = comm.rank()
    g, ocfg, params, head_w, head_b = load_case("bert_small_b2_s16")
    if rank == 1:     # rank 1 starts from different weights: the step-0 broadcast must fix that
        params = {k: v + 0.01 for k, v in params.items()}
    model = build_model(ocfg, params, head_w, head_b, "f32")
    steps = 3
    epochs = 2 if mode.endswith("_epochs2") else 1     # the second epoch re-broadcasts weights AND optimizer variables from rank 0
    opt = AdamWeightDecay(learning_rate=warmup_scheduler(steps * epochs, 1e-3), weight_decay_rate=0.01)
    validate = mode == "allreduce_epochs2"
    from polus_amd.metrics import Accuracy
    trainer = ClassifierTrainer(model, opt, SparseCategoricalCrossentropy(), metrics=[Accuracy(4)] if validate else [])
    assert trainer.use_horovod
    accum = 2 if mode.endswith("_accum2") else 1
    trainer.grad_accum_steps = accum
    # several buckets even for this small model; `_split`: buckets far smaller than most tensors, so nearly every tensor is cut
    # across buckets and updated part by part (what happens to the 89 MB word-embedding gradient of BERT-base at 64 MB)
    os.environ["POLUS_BUCKET_MB"] = "0.004" if mode.endswith("_split") else "0.05"
    mine = list(shard(range(4), 2, rank))                  # sample i -> rank i mod 2
    batches = []
    for s in range(steps):
        ids, mask, tt, labels = synth_batch(ocfg, 4, 16, 4, 420 + s)
        for part in ([mine] if accum == 1 else [[m] for m in mine]):       # accumulation: one sample per micro-step
            batches.append(({"input_ids": ids[part], "attention_mask": mask[part], "token_type_ids": tt[part]},
                            labels[part]))
    calls = []
    orig = opt.apply_gradients

    def counted(gv, **kw):
        gv = list(gv)
        if kw.get("_ranges") and scheme == "allreduce":       # a part of ONE tensor that is cut across buckets
            (v,) = [v for _, v in gv]
            n = sum(max(0, min(hi, v.offset + v.size) - max(lo, v.offset)) for lo, hi in kw["_ranges"])
        else:
            n = sum(v.size for _, v in gv)
        calls.append((n, kw))
        return orig(gv, **kw)
    opt.apply_gradients = counted
    # the CU reserve for RCCL's channel kernels is on exactly around a backward pass that carries the exchange
    from polus_amd import ops
    toggles, orig_reserve = [], ops.reserve_cus
    ops.reserve_cus = lambda on: (toggles.append(bool(on)), orig_reserve(on))[1]
    callbacks = []
    if validate:
        # polus/callbacks.py:218-261 under data parallelism: every rank predicts its own validation shard on the GPU,
        # hvd.allgather_object collects the (prediction, label) pairs and rank 0 feeds its metrics
        import torch
        from polus_amd.callbacks import ValidationDataCallback
        val = []
        for s in range(2):
            ids, mask, tt, labels = synth_batch(ocfg, 4, 16, 4, 900 + s)
            val.append(({"input_ids": torch.from_numpy(ids[mine]).cuda(), "attention_mask": torch.from_numpy(mask[mine]).cuda(),
                         "token_type_ids": torch.from_numpy(tt[mine]).cuda()}, torch.from_numpy(labels[mine]).cuda()))
        vcb = ValidationDataCallback(val, name="val")
        callbacks = [vcb]
    trainer.train(batches, epochs=epochs, callbacks=callbacks)
    if validate:
        # what the callback reported after the LAST epoch == accuracy over BOTH ranks' shards with the final weights
        hits = total = 0
        for x, y in val:
            pred = model.inference(x)
            hits += int((pred == y.to(pred.dtype)).sum()); total += y.numel()
        both = comm.allgather_object((hits, total))
        expect = sum(h for h, _ in both) / sum(t for _, t in both)
        if rank == 0:
            got = vcb.get_metrics()["Accuracy"]
            assert len(got) == epochs and abs(got[-1] - expect) < 1e-12, (got, expect)
            assert both[0] != both[1], "the two validation shards should differ"
    assert trainer._dp_mode() == scheme
    steps *= epochs
    assert toggles == [True, False] * steps, toggles
    if scheme == "allreduce":
        # the update of every bucket is queued behind its all-reduce: one launch per bucket and step, every
        # variable exactly once per step, the step counter advanced by the first launch only
        r = trainer._reducer(model.arena)
        nb = len(r.buckets)
        total = sum(v.size for v in trainer.trainable_weights)
        assert nb > 1 and len(calls) % steps == 0 and len(calls) >= steps, (nb, len(calls))   # buckets of arena padding hold no tensor
        per = len(calls) // steps
        if mode.endswith("_split"):
            assert any(c[1].get("_ranges") for c in calls) and nb > len(trainer.trainable_weights), "no tensor was cut across buckets"
        for k in range(steps):
            per_step = calls[per * k:per * (k + 1)]
            assert sum(c[0] for c in per_step) == total, "every parameter exactly once per step"
            assert per_step[0][1].get("_advance") is True and all(c[1].get("_advance") is False for c in per_step[1:])
        assert opt.iterations == steps
    else:
        # one launch per optimizer step, on the windows this rank owns after the reduce-scatter
        assert len(calls) == steps and all(c[1].get("_ranges") for c in calls), calls
        r = trainer._reducer(model.arena)
        assert len(r.buckets) > 1 and sum(hi - lo for lo, hi in r.owned_ranges()) == model.arena.grads.numel() // 2
        assert opt.iterations == steps
    # the parent compares parameters, which depend on every step's averaged gradients
    flat = model.arena.params.detach().float().cpu().numpy()
    np.save(f"{out}.rank{rank}.npy", flat)
    if mode == "rs_epochs2":
        # the moments are sharded after the last step; saving them unsynchronised is refused, the collective sync
        # makes them whole and identical on both ranks (the parent compares them with the single-process run)
        import pytest
        from polus_amd.checkpoint import save_training_state
        with pytest.raises(RuntimeError, match="sync_optimizer_state"):
            save_training_state(trainer, out + f".rank{rank}")
        trainer.sync_optimizer_state()
        m_, v_ = opt._slots(model.arena)
        np.save(f"{out}.rank{rank}.m.npy", m_.detach().float().cpu().numpy())
        np.save(f"{out}.rank{rank}.v.npy", v_.detach().float().cpu().numpy())
        save_training_state(trainer, out + f".rank{rank}")
    comm.barrier()
    comm.shutdown()
    print(f"rank {rank} OK", flush=True)


if __name__ == "__main__":
    main()
