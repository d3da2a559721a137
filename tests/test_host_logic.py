"""CPU: host-side logic of the drop-in API — trainer loop and hook order, callbacks, metrics,
schedules, sharding, the bucket planner — exercised with NumPy test doubles (tests/fakes.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import optim as oo
from polus_amd import comm
from polus_amd.callbacks import (Callback, CallbackCoordinator, ConsoleLogCallback, EarlyStop, ICallback, IOutput,
                                 LossSmoothCallback, Profiler, SaveModelCallback, TimerCallback,
                                 ValidationDataCallback)
from polus_amd.data import ShardedBatches, shard
from polus_amd.metrics import IMetric, IConfusionMatrixTF, MacroF1Score
from polus_amd.optimizers import WarmUpLinearDecay, _LearningRate, default_no_decay
from polus_amd.schedulers import warmup_scheduler
from polus_amd.tensor import DeviceScalar
from polus_amd.training import BaseTrainer, ClassifierTrainer, Trainer
from tests.fakes import FakeLinearModel, FakeSGD, FakeXent


def make_data(n_batches=3, bs=4, seed=0):
    r = np.random.default_rng(seed)
    return [(r.standard_normal((bs, 5)), r.integers(0, 3, size=bs)) for _ in range(n_batches)]


class Recorder(Callback):
    def __init__(self, log):
        super().__init__()
        self.log = log

    def on_train_begin(self): self.log.append("train_begin")
    def on_epoch_begin(self, e): self.log.append(f"epoch_begin{e}")
    def on_train_batch_begin(self, e, s): self.log.append(f"batch_begin{e}.{s}")
    def on_train_batch_end(self, e, s, loss): self.log.append(f"batch_end{e}.{s}")
    def on_epoch_end(self, e): self.log.append(f"epoch_end{e}")
    def on_train_end(self): self.log.append("train_end")


def test_base_trainer_is_abstract_and_hooks_default():
    with pytest.raises(Exception):
        BaseTrainer(FakeLinearModel(), FakeSGD(), FakeXent())

    class T(BaseTrainer):
        pass
    t = T(FakeLinearModel(), FakeSGD(), FakeXent())
    assert t.forward_without_grads(1, 2) == (1, 2)
    with pytest.raises(NotImplementedError):
        t.forward_with_grads(1)
    assert Trainer is ClassifierTrainer
    assert t.use_horovod is False and t.step_counter == 0 and t.early_stop is False


def test_train_loop_hook_order_matches_reference():
    """polus/training.py:299-338: batch_begin fires before the fetch, once extra at exhaustion."""
    log = []
    trainer = ClassifierTrainer(FakeLinearModel(), FakeSGD(), FakeXent())
    trainer.train(make_data(2), epochs=2, callbacks=[Recorder(log)])
    assert log == ["train_begin",
                   "epoch_begin0", "batch_begin0.0", "batch_end0.0", "batch_begin0.1", "batch_end0.1", "batch_begin0.2", "epoch_end0",
                   "epoch_begin1", "batch_begin1.0", "batch_end1.0", "batch_begin1.1", "batch_end1.1", "batch_begin1.2", "epoch_end1",
                   "train_end"]
    assert trainer.step_counter == 4
    assert trainer.callbacks.steps == 2 and trainer.callbacks.epochs == 2


def test_train_argument_fallbacks_and_errors():
    trainer = ClassifierTrainer(FakeLinearModel(), FakeSGD(), FakeXent())
    with pytest.raises(ValueError):
        trainer.train()
    with pytest.raises(ValueError):
        trainer.train(make_data(1))
    log = []
    seen = []
    trainer.changing_train_config(tf_dataset=make_data(1), epochs=1, callbacks=[Recorder(log)],
                                  train_map_f=lambda d: (seen.append(1), d)[1])
    trainer.train()
    assert "batch_end0.0" in log and seen == [1]
    # legacy kwarg (polus/training.py:275-276)
    seen2 = []
    trainer.train(custom_data_transform_f=lambda d: (seen2.append(1), d)[1])
    assert seen2 == [1]


def test_train_step_order_and_sgd_math():
    model = FakeLinearModel()
    order = []

    class T(ClassifierTrainer):
        def forward_without_grads(self, *a):
            order.append("without"); return a

        def forward_with_grads(self, x, y):
            order.append("with"); return super().forward_with_grads(x, y)
    post = []
    trainer = T(model, FakeSGD(0.5), FakeXent(), post_process_logits=lambda l: (post.append("logits"), l)[1],
                post_process_grads=lambda g: (post.append("grads"), g)[1])
    x, y = make_data(1)[0]
    w0 = model.w.value.clone()
    loss = trainer.train_step(x, y)
    assert order == ["without", "with"] and post == ["logits", "grads"]
    assert model.calls[0] == ("call", True)
    assert torch.allclose(model.w.value, w0 - 0.5 * model.w.grad)
    assert isinstance(loss, float)


def test_gradient_accumulation_steps():
    m1, m2 = FakeLinearModel(), FakeLinearModel()
    data = make_data(2, bs=4)
    t1 = ClassifierTrainer(m1, FakeSGD(1.0), FakeXent())
    t1.grad_accum_steps = 2
    t1.train_step(*data[0])
    assert torch.equal(m1.w.value, m2.w.value)           # no update on the first micro-step
    t1.train_step(*data[1])
    # equals one step on the concatenated batch (mean of means, equal sizes)
    t2 = ClassifierTrainer(m2, FakeSGD(1.0), FakeXent())
    t2.train_step(np.concatenate([data[0][0], data[1][0]]), np.concatenate([data[0][1], data[1][1]]))
    assert torch.allclose(m1.w.value, m2.w.value, atol=1e-6)


def test_early_stop_flag_breaks_loops():
    class Stopper(Callback):
        def on_train_batch_end(self, e, s, loss):
            self.coordinator.trainer.early_stop = True
    log = []
    trainer = ClassifierTrainer(FakeLinearModel(), FakeSGD(), FakeXent())
    trainer.train(make_data(3), epochs=3, callbacks=[Stopper(), Recorder(log)])
    assert log.count("batch_end0.0") == 1 and "batch_end0.1" not in log and "epoch_begin1" not in log
    assert log[-2:] == ["epoch_end0", "train_end"]


def test_callbacks_bus_and_stock_callbacks(tmp_path, capsys):
    with pytest.raises(Exception):
        ICallback()
    model = FakeLinearModel()
    trainer = ClassifierTrainer(model, FakeSGD(), FakeXent(), metrics=[MacroF1Score(num_classes=3)])
    val = make_data(2)
    console = ConsoleLogCallback()
    smooth = LossSmoothCallback(output=True)
    cbs = [smooth, TimerCallback(), ValidationDataCallback(val, name="val"), console, EarlyStop(),
           SaveModelCallback("every", cache_folder=str(tmp_path)), SaveModelCallback("end")]
    trainer.train(make_data(3), epochs=2, callbacks=cbs)
    sd = trainer.callbacks.shared_dict
    assert set(sd) >= {"smooth_loss", "validation"}
    assert len(sd["validation"]["val"]["MacroF1Score"]) == 2
    assert all(0.0 <= v <= 1.0 for v in sd["validation"]["val"]["MacroF1Score"])
    assert trainer.callbacks.output_streamers == [console]
    assert trainer.callbacks.has_callback(TimerCallback) and not trainer.callbacks.has_callback(Profiler)
    saves = [c for c in model.calls if c[0] == "save"]
    assert [s[1].get("extension") for s in saves] == ["_epoch_0", "_epoch_1", None]
    assert saves[0][1]["base_path"] == str(tmp_path)
    # smooth loss = bias-corrected EMA (polus/callbacks.py:176-182)
    assert smooth.n == 6 and smooth.smooth_loss > 0
    out = capsys.readouterr().out
    assert "loss:" in out and "smooth loss" in out and "time" in out


def test_save_best_and_earlystop_quirks():
    class M:  # minimal coordinator stand-in
        pass
    cb = SaveModelCallback("best", validation_name="v", metric_name="m")
    assert cb.best == 0
    model = FakeLinearModel()
    coord = CallbackCoordinator([cb], trainer=type("T", (), {"model": model})(), epochs=1, steps=1)
    coord.shared_dict["validation"] = {"v": {"m": [0.4]}}
    cb.on_epoch_end(0)
    coord.shared_dict["validation"]["v"]["m"].append(0.3)
    cb.on_epoch_end(1)
    assert [c[1]["extension"] for c in model.calls if c[0] == "save"] == ["_v_m_best"] and cb.best == 0.4
    es = EarlyStop(patience=0)
    tr = type("T", (), {"model": model, "early_stop": False})()
    coord = CallbackCoordinator([es], trainer=tr, epochs=1, steps=1)
    es.on_train_begin()            # no LossSmoothCallback -> falls back to the raw loss
    assert es.use_smooth_loss is False
    es.on_train_batch_end(0, 0, 5.0); es.on_epoch_end(0)
    assert tr.early_stop is False and es.last_loss == 1000     # quirk: never updated
    es.on_train_batch_end(0, 0, float("nan")); es.on_epoch_end(1)
    assert tr.early_stop is True


def test_profiler_env_adds_callback(monkeypatch):
    """polus/training.py:279-285: POLUS_PROFILER appends a Profiler with POLUS_PROFILER_RANGE."""
    monkeypatch.setenv("POLUS_PROFILER", "yes")
    monkeypatch.setenv("POLUS_PROFILER_RANGE", "100:200")      # never reached: no HIP events on CPU
    trainer = ClassifierTrainer(FakeLinearModel(), FakeSGD(), FakeXent())
    trainer.train(make_data(1), epochs=1, callbacks=[])
    profs = [c for c in trainer.callbacks.callbacks if isinstance(c, Profiler)]
    assert len(profs) == 1 and profs[0].steps_interval == [100, 200]


def test_metrics_protocol_and_macro_f1():
    with pytest.raises(Exception):
        IMetric()
    with pytest.raises(Exception):
        IConfusionMatrixTF(3)
    m = MacroF1Score(num_classes=3)
    assert m.name == "MacroF1Score"
    m.samples_from_batch((np.array([0, 1, 2, 2, 1]), np.array([0, 2, 2, 2, 1])))
    m.samples_from_batch((torch.tensor([0]), torch.tensor([0])))
    cm = oo.confusion_matrix([0, 1, 2, 2, 1, 0], [0, 2, 2, 2, 1, 0], 3)
    assert np.array_equal(m.confusion_matrix, cm) and m.confusion_matrix.dtype == np.int32
    assert abs(m.evaluate() - oo.macro_f1(cm)) < 1e-15
    assert m.confusion_matrix.sum() == 0                      # evaluate() resets
    m2 = MacroF1Score(num_classes=2, reduce_f=lambda s: (s[0][:1], s[1][:1]))
    m2.samples_from_batch((np.array([1, 0]), np.array([1, 1])))
    assert m2.confusion_matrix.sum() == 1


def test_schedule_and_lr_handle():
    s = warmup_scheduler(100, 1e-3, end_lr=5e-5)              # end_lr ignored, as in the reference
    assert isinstance(s, WarmUpLinearDecay)
    for t in [0, 3, 9, 10, 11, 55, 99, 100, 101, 1000]:
        assert abs(s(t) - oo.warmup_linear_lr(t, 100, 1e-3)) < 1e-18
    lr = _LearningRate(0.01)
    assert lr.read_value() == 0.01
    lr.scale(8)
    assert abs(lr(0) - 0.08) < 1e-12
    lr2 = _LearningRate(s)
    lr2.scale(2)
    assert abs(lr2(5) - 2 * s(5)) < 1e-18
    assert default_no_decay("layer3.ln1.g") and default_no_decay("layer0.qkv.b") and default_no_decay("emb.ln.b")
    assert not default_no_decay("layer0.qkv.w") and not default_no_decay("emb.word")
    assert default_no_decay("x.b") == oo.is_no_decay("x.b")


def test_device_scalar_behaves_like_a_number():
    d = DeviceScalar(torch.tensor([1.23456]))
    assert f"{d:.3f}" == "1.235" and abs(float(d) - 1.23456) < 1e-6
    assert abs((0.5 * d) - 0.61728) < 1e-5 and abs(sum([d, d]) - 2.46912) < 1e-5
    assert d > 1 and d < 2 and not math.isnan(d.item())


def test_shard_rule_and_batches():
    assert list(shard(range(10), 4, 1)) == oo.shard_indices(10, 4, 1)
    sb = ShardedBatches(lambda: iter(range(11)), 2, lambda b: list(b), drop_remainder=True, num_shards=2, index=0)
    assert list(sb) == [[0, 2], [4, 6], [8, 10]]
    sb = ShardedBatches(lambda: iter(range(11)), 2, lambda b: list(b), drop_remainder=False, num_shards=2, index=1)
    assert list(sb) == [[1, 3], [5, 7], [9]]
    assert list(sb) == [[1, 3], [5, 7], [9]]                 # re-iterable


def test_bucket_planner_covers_arena_in_reverse_order():
    g = torch.zeros(1000)
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 300, boundaries=[0, 100, 250, 600, 900])
    assert r.buckets == [(900, 1000), (600, 900), (250, 600), (0, 250)]   # <= 300 elements unless one tensor is larger
    assert sum(hi - lo for lo, hi in r.buckets) == 1000
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 50, boundaries=[0, 100, 250, 600, 900])
    assert r.buckets == [(900, 1000), (600, 900), (250, 600), (100, 250), (0, 100)]   # a tensor is never split
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 256)
    assert r.buckets[0][1] == 1000 and r.buckets[-1][0] == 0
    assert all(r.buckets[i][0] == r.buckets[i + 1][1] for i in range(len(r.buckets) - 1))


def test_mock_comm_surface_world_size_one():
    """polus/mock/horovod.py:5-24."""
    assert comm.size() == 1 and comm.local_rank() == 0 and comm.rank() == 0
    assert comm.allgather_object(("a", 1)) == [("a", 1)]
    assert comm.broadcast_variables([torch.zeros(2)]) is None
    tape = object()
    assert comm.DistributedGradientTape(tape) is tape
    assert comm.init() in ("mock", "gloo", "nccl")


def test_in_backward_update_gating_and_state_sync_on_the_host():
    """The optional step schedules fall back cleanly where they do not apply: no in-backward optimizer update for an
    arena that is not on the GPU or an optimizer that is not the fused Adam (the single launch after backward stays);
    trainer.sync_optimizer_state() is a no-op in a single process; save_training_state refuses Adam moments that a
    data-parallel reduce-scatter step left current on this rank's slices only."""
    from polus_amd.checkpoint import save_training_state
    model = FakeLinearModel(seed=3)
    trainer = ClassifierTrainer(model, FakeSGD(0.1), FakeXent())
    assert not trainer.use_horovod
    assert trainer._updater() is None                     # CPU arena, SGD test double
    trainer.update_in_backward = False
    assert trainer._updater() is None
    trainer.update_in_backward = True
    x, y = make_data(1)[0]
    before = model.arena.params.clone()
    trainer.train_step(x, y)                              # takes the plain path: one apply_gradients after backward
    assert not torch.equal(before, model.arena.params) and model.grad_ready_hook is None
    trainer.sync_optimizer_state()                        # nothing to gather in one process
    assert getattr(trainer, "_opt_state_synced", True)
    # the guard of the resume state: a (simulated) data-parallel trainer whose moments are sharded
    trainer.use_horovod, trainer._opt_state_synced = True, False
    with pytest.raises(RuntimeError, match="sync_optimizer_state"):
        save_training_state(trainer, "/nonexistent/never-written")
    trainer.use_horovod, trainer._opt_state_synced = False, True


def test_gemm_split_auto_is_declared_and_bucket_callback_fires_in_launch_order():
    """polus_gemm_auto_split is part of the C ABI the host binds; GradBucketReducer.on_launched hands every bucket,
    in launch (descending-offset) order, to the trainer that queues the bucket's update behind it."""
    from polus_amd import _lib
    assert "polus_gemm_auto_split" in _lib.SIGNATURES
    g = torch.arange(1000, dtype=torch.float32)
    r = comm.GradBucketReducer(g, bucket_bytes=4 * 300, boundaries=[0, 100, 250, 600, 900], plane=_EchoPlane())
    seen = []
    r.on_launched = lambda lo, hi, work: seen.append((lo, hi, work))
    r.begin()
    r.on_ready(900, 1000)
    assert [(lo, hi) for lo, hi, _ in seen] == [(900, 1000)]
    r.on_ready(250, 900)
    r.finish()
    assert [(lo, hi) for lo, hi, _ in seen] == list(r.buckets) and seen[0][0] > seen[-1][0]
    assert all(w.waited >= 1 for _, _, w in seen)         # finish() waited for every bucket


class _EchoWork:
    def __init__(self):
        self.waited = 0

    def wait(self):
        self.waited += 1


class _EchoPlane:
    """A data plane that reduces nothing (world size 1): records the calls."""
    def all_reduce_sum(self, t):
        return _EchoWork()


@pytest.mark.parametrize("bucket_mb", [64, 16, 32, 128])
def test_bucket_plan_of_bert_base_at_world_8(monkeypatch, bucket_mb):
    """8-GPU readiness that needs no hardware: the bucket plans of the BERT-base gradient arena at world size 8, for the
    default bucket size and the two others trainer.tune_data_parallel tries.
    `rs`: every bucket a multiple of 64 x 8 elements (whole 256-byte lines per rank), the arena covered exactly once.
    `allreduce`: cut at tensor boundaries, and a tensor larger than a bucket (the 89 MB word-embedding gradient) cut into
    parts no larger than a bucket, so that no single collective + optimizer sweep of that size ends the step."""
    import torch
    from polus_amd import comm
    monkeypatch.setitem(comm._STATE, "world", 8)
    monkeypatch.setitem(comm._STATE, "rank", 3)
    H, I, L, V = 768, 3072, 12, 28996
    sizes = [V * H, 512 * H, 2 * H, H, H]
    for _ in range(L):
        sizes += [3 * H * H, 3 * H, H * H, H, H, H, I * H, I, H * I, H, H, H]
    sizes += [4 * H, 4]
    offs, n = [], 0
    for sz in sizes:
        offs.append(n)
        n += (sz + 63) // 64 * 64
    n = (n + 64 * 1680 - 1) // (64 * 1680) * (64 * 1680)        # ParamArena's world-independent padding
    grads = torch.empty(n, dtype=torch.float32, device="meta")
    elems = (bucket_mb << 20) // 4
    for mode in ("rs", "allreduce"):
        r = comm.GradBucketReducer(grads, bucket_bytes=bucket_mb << 20, boundaries=offs, mode=mode, plane=object())
        b = r.buckets
        assert b[0][1] == n and b[-1][0] == 0 and all(lo < hi for lo, hi in b)
        assert all(b[k][0] == b[k + 1][1] for k in range(len(b) - 1)), "buckets must tile the arena from its end"
        if mode == "rs":
            assert all((hi - lo) % (64 * 8) == 0 for lo, hi in b)
            own = r.owned_ranges()
            assert sum(hi - lo for lo, hi in own) * 8 == n
        else:
            assert max(hi - lo for lo, hi in b) <= elems + 512, "no bucket beyond its size: the embedding table is split"
            emb = [(lo, hi) for lo, hi in b if lo < V * H]
            assert (len(emb) >= 2 or V * H <= elems) and all(lo % 64 == 0 for lo, _ in b)
