"""Trainer drop-in for polus/training.py: BaseTrainer / ClassifierTrainer with the same
constructor, attributes, hooks and training-loop order, over the MI355X engine.

What changes underneath ``train_step`` (polus/training.py:150-193): the GradientTape is
replaced by the models' explicit backward (HIP kernels), ``hvd.DistributedGradientTape`` by
a bucketed RCCL all-reduce that starts while backward is still running, and
``optimizer.apply_gradients`` by one fused multi-tensor launch.
"""
import os

from . import _lib, comm
from .optimizers import Adam
from .callbacks import CallbackCoordinator, Profiler
from .context import PolusContext, logger

hvd = comm


class _UpdateInBackward:
    """Single-process steps: the fused optimizer update of a gradient window starts as soon as backward reports the
    window final (`model.grad_ready_hook`), on its own stream, beside the backward pass of the layers below -- the
    update is HBM-bound, backward mostly MFMA-bound.  Same kernel, same arithmetic per parameter as the one launch
    of `optimizer.apply_gradients` after backward (polus/training.py:191): the parameters come out bit-identical
    (tests/test_boundary_gpu.py).  The transposed shadows of a window's GEMM weights are re-derived right behind its
    update; the main stream joins before `train_step` returns."""

    def __init__(self, trainer, arena):
        import torch
        self.trainer, self.arena = trainer, arena
        self.stream = torch.cuda.Stream(device=arena.device)
        self.fork, self.done = torch.cuda.Event(), torch.cuda.Event()
        self.vars = sorted(trainer.trainable_weights, key=lambda v: v.offset)

    def begin(self):
        self.first = True
        self.left = {id(v): v for v in self.vars}

    def _apply(self, vs):
        import torch
        main = torch.cuda.current_stream()
        self.fork.record(main)               # everything that read these parameters was launched before this point
        self.stream.wait_event(self.fork)
        with _lib.stream_scope(self.stream):
            self.trainer.optimizer.apply_gradients([(v.grad, v) for v in vs], _advance=self.first, _refresh=False)
            self.arena.refresh_transposed_of(vs)     # the dX GEMMs that read these W^T completed before the fork event
        self.first = False
        for v in vs:
            del self.left[id(v)]

    def on_ready(self, lo, hi, variables=None):
        # by identity: a model may report windows of an arena this trainer does not update
        vs = [v for v in (variables or ()) if id(v) in self.left]
        if vs:
            self._apply(vs)

    def finish(self):
        import torch
        if self.left:                        # windows backward never reported (a model without the hook calls)
            self._apply(list(self.left.values()))
        with _lib.stream_scope(self.stream):
            self.done.record(self.stream)
        torch.cuda.current_stream().wait_event(self.done)


class _UpdateBehindAllReduce(_UpdateInBackward):
    """Data-parallel steps on the all-reduce scheme: the reducer fires a bucket's all-reduce as soon as backward
    has produced everything in it; the fused optimizer update of the variables inside that bucket is queued on the
    update stream right behind the all-reduce's completion event, so it too runs beside the rest of backward.
    Buckets are cut at variable boundaries, so every variable belongs to exactly one of them.  When the all-reduce
    has completed, the layers that read these parameters have completed as well (the collective waited for the
    compute stream) -- no further fence is needed.  What stays exposed after backward is the last bucket (the
    embeddings) and its update."""

    def begin(self):
        super().begin()
        self.part_left = {}                  # id(v) -> elements of a tensor cut across buckets that are still to be updated

    def on_bucket(self, lo, hi, work):
        vs = [v for v in self.vars if id(v) in self.left and lo <= v.offset and v.offset + v.size <= hi]
        # tensors larger than a bucket (the word-embedding table) are cut across several: update the part inside this one
        parts = [v for v in self.vars if id(v) in self.left and v.offset < hi and v.offset + v.size > lo and
                 not (lo <= v.offset and v.offset + v.size <= hi)]
        opt = self.trainer.optimizer
        with _lib.stream_scope(self.stream):
            work.wait()                      # the update stream (not the compute stream) waits for this bucket
            if vs:
                opt.apply_gradients([(v.grad, v) for v in vs], _advance=self.first, _refresh=False)
                self.arena.refresh_transposed_of(vs)
                self.first = False
            for v in parts:
                a, b = max(lo, v.offset), min(hi, v.offset + v.size)
                opt.apply_gradients([(v.grad, v)], _advance=self.first, _refresh=False, _ranges=[(a, b)])
                self.first = False
                self.part_left[id(v)] = self.part_left.get(id(v), v.size) - (b - a)
                if self.part_left[id(v)] <= 0:
                    self.arena.refresh_transposed_of([v])
                    del self.left[id(v)]
        for v in vs:
            del self.left[id(v)]


class BaseTrainer:
    """polus/training.py:14-338."""

    def __init__(self, model, optimizer, loss, metrics=[], post_process_logits=None, post_process_grads=None):
        if self.__class__.__name__ == "BaseTrainer":
            raise Exception("This is an abstraction that cannot be instantiated")
        super().__init__()
        self.model = model
        self.loss = loss
        self.optimizer = optimizer
        self.post_process_logits = post_process_logits
        self.post_process_grads = post_process_grads
        self.metrics = metrics
        self.early_stop = False
        self.train_config = {}
        self.step_counter = 0
        self.grad_accum_steps = 1

        if not hasattr(self, "trainable_weights"):
            logger.warning(f"Since no specific trainable_weights were defined during the {self.__class__.__name__} "
                           "instantiation, the trainer will optimizer all the variables found on the model instance")
            self.trainable_weights = model.trainable_weights

        self.use_horovod = PolusContext().is_horovod_enabled()
        self._reducers = {}
        if self.use_horovod:
            # polus/training.py:90-94: learning rate x world size
            if hasattr(optimizer, "learning_rate"):
                optimizer.learning_rate.scale(hvd.size())
                logger.info("The learning rate was adjusted to account for the multiGPU training")
            else:
                logger.info("It was not possible to change the learning rate for the multiGPU training, "
                            "please multiply the learning rate by hvd.size()")

    def __str__(self):
        return "Trainer"

    # ---- hooks (same contract as the reference)
    def forward_without_grads(self, *inputs):
        return inputs

    def forward_with_grads(self, *inputs):
        raise NotImplementedError("forward_with_grads function must be implemented in order to compute a loss "
                                  "value for optimization")

    def backward_from_loss(self, accumulate=False):
        """The explicit counterpart of tape.gradient (polus/training.py:185): pull the
        gradient of the loss wrt the model output and push it through the model."""
        dlogits = self.loss.backward(accumulate)
        self.model.backward(dlogits, accumulate=accumulate)

    # ---- gradient exchange
    def _arenas(self):
        seen, out = set(), []
        for v in self.trainable_weights:
            if id(v.arena) not in seen:
                seen.add(id(v.arena))
                out.append(v.arena)
        return out

    def _dp_mode(self):
        """How the gradients of this trainer are exchanged (decided once).  Default "allreduce": bucketed
        all-reduce fired inside backward, each bucket's fused AdamW queued behind it (_UpdateBehindAllReduce) --
        everything but the last bucket hides under backward.  POLUS_DP_MODE=rs selects reduce-scatter -> AdamW
        on the owned slices -> all-gather of the parameters (half the bytes during backward and 1/N of the
        optimizer traffic, but the all-gather of the f32 parameters is exposed after the update: (N-1)/N x
        438 MB over the links, ~2.9 ms at N = 2), when one fused optimizer sweeps one arena and nothing
        needs every gradient on every rank (no post_process_grads, no global-norm clipping)."""
        m = getattr(self, "_dp_mode_cached", None)
        if m is None:
            arenas = self._arenas()
            want = os.environ.get("POLUS_DP_MODE", "allreduce") == "rs"
            ok = (len(arenas) == 1 and isinstance(self.optimizer, Adam) and not self.optimizer.global_clipnorm and
                  self.post_process_grads is None and hasattr(arenas[0], "size") and
                  arenas[0].grads.numel() % (64 * hvd.size()) == 0 and
                  all(v.arena is arenas[0] for v in self.trainable_weights) and want)
            if want and not ok:
                logger.warning("POLUS_DP_MODE=rs was requested but this trainer cannot use it (it needs ONE arena swept by the "
                               "fused Adam, no post_process_grads / global_clipnorm, and an arena of a multiple of "
                               f"64 x {hvd.size()} elements): exchanging gradients by all-reduce instead")
            m = self._dp_mode_cached = "rs" if ok else "allreduce"
        return m

    def _updater(self):
        """The in-backward optimizer update (see _UpdateInBackward), when nothing needs all gradients at once: one
        process, one arena on the GPU swept by the fused Adam, no post_process_grads, no global-norm clipping, no
        step graph (its eager warm-up steps must take the path that is captured), and -- unless POLUS_UPDATE_IN_BACKWARD=1 --
        at least 6144 tokens per step (below, the windows outlast the launches they hide under).  POLUS_UPDATE_IN_BACKWARD=0 keeps
        the single launch after backward; `trainer.update_in_backward = False` does the same for the steps that
        follow (bench.py's instrumented single-stream step)."""
        if getattr(self, "_graphed", None) is not None or not getattr(self, "update_in_backward", True):
            return None
        key = tuple(id(v) for v in self.trainable_weights)
        u = getattr(self, "_updater_cached", False)
        if u is False or getattr(self, "_updater_key", None) != key:
            self._updater_key = key
            arenas = self._arenas()
            ok = (os.environ.get("POLUS_UPDATE_IN_BACKWARD", "1") != "0" and len(arenas) == 1 and
                  hasattr(self.model, "grad_ready_hook") and isinstance(self.optimizer, Adam) and
                  hasattr(self.optimizer, "grad_scale") and not self.optimizer.global_clipnorm and
                  self.post_process_grads is None and hasattr(arenas[0], "refresh_transposed") and
                  arenas[0].grads.is_cuda and all(v.arena is arenas[0] for v in self.trainable_weights))
            u = self._updater_cached = _UpdateInBackward(self, arenas[0]) if ok else None
        if u is not None and "POLUS_UPDATE_IN_BACKWARD" not in os.environ:
            # By size, unless the switch says 0 / 1: a layer's update window (its parameters' 213 MB for BERT-base, whatever the
            # batch) rides on the CUs the weight-gradient launch of the layer below leaves free, and that launch shrinks with the
            # tokens per step: 247 us at 16384 tokens, 87 us at 4096 -- there the window outlasts it and lands on the next dU GEMM
            # (BASELINE configs[1], B=32 S=128: 5.56 -> 5.44 ms per step with the one launch after backward; neutral at 8192 tokens;
            # at 16384 the in-backward form wins, 12.48 -> 12.27 ms in round 2).
            tokens = getattr(self.model, "tokens_per_step", None)
            if tokens is not None and tokens < 6144:
                return None
        return u

    def _updater_dp(self, reducer):
        """The per-bucket optimizer update behind each all-reduce (_UpdateBehindAllReduce): all-reduce scheme, GPU
        arena, the fused Adam, nothing that needs every gradient at once.  POLUS_UPDATE_IN_BACKWARD=0 keeps the
        update after backward (split around the last bucket)."""
        if self._dp_mode() != "allreduce" or reducer.mode == "rs" or not getattr(self, "update_in_backward", True):
            return None
        key = tuple(id(v) for v in self.trainable_weights)
        u = getattr(self, "_updater_dp_cached", False)
        if u is False or getattr(self, "_updater_dp_key", None) != key:
            self._updater_dp_key = key
            arena = self._arenas()[0]
            ok = (os.environ.get("POLUS_UPDATE_IN_BACKWARD", "1") != "0" and isinstance(self.optimizer, Adam) and
                  hasattr(self.optimizer, "grad_scale") and not self.optimizer.global_clipnorm and
                  self.post_process_grads is None and hasattr(arena, "refresh_transposed") and arena.grads.is_cuda and
                  all(v.arena is arena for v in self.trainable_weights) and getattr(self, "_graphed", None) is None)
            u = self._updater_dp_cached = _UpdateBehindAllReduce(self, arena) if ok else None
        return u

    def _reducer(self, arena):
        r = self._reducers.get(id(arena))
        if r is None:
            import torch
            bucket = int(float(getattr(self, "bucket_mb", None) or os.environ.get("POLUS_BUCKET_MB", "64")) * (1 << 20))
            mode = self._dp_mode()
            bf16 = os.environ.get("POLUS_DP_BF16", "0") == "1" and arena.grads.is_cuda
            r = comm.GradBucketReducer(arena.grads, bucket_bytes=bucket, boundaries=[v.offset for v in arena.vars],
                                       mode=mode, transport_dtype=torch.bfloat16 if bf16 else None)
            self._reducers[id(arena)] = r
        return r

    def train_step(self, *inputs):
        """polus/training.py:150-193, same order: forward_without_grads -> forward_with_grads
        -> loss -> gradients (all-reduced across ranks) -> post_process_grads -> apply."""
        g = getattr(self, "_graphed", None)
        if g is not None:
            return g(*inputs)
        return self._eager_step(*inputs)

    def _eager_step(self, *inputs):
        with _lib.pinned_stream():      # one stream lookup per step instead of one per launch
            return self._train_step(*inputs)

    def enable_step_graph(self, warmup=3):
        """Replay the step from a captured HIP graph after `warmup` eager steps (polus_amd/graph.py): for
        launch-bound workloads.  Same results bit for bit; single process, static batch shape."""
        from .graph import GraphedStep
        self._graphed = GraphedStep(self, warmup)
        return self

    def disable_step_graph(self):
        self._graphed = None

    def _train_step(self, *inputs):
        micro = self.step_counter_micro = getattr(self, "step_counter_micro", 0)
        accum = self.grad_accum_steps
        first, last = (micro % accum == 0), (micro % accum == accum - 1)

        inputs = self.forward_without_grads(*inputs)
        inputs = self.forward_with_grads(*inputs)
        loss_value = self.loss(*inputs)

        # The exchange starts inside backward: on the micro-step that completes an accumulation the model
        # reports each gradient window as it becomes final and the reducer fires the buckets above it.
        reducers, dp_updater = [], None
        if self.use_horovod and last:
            arenas = self._arenas()
            if len(arenas) == 1 and hasattr(self.model, "grad_ready_hook"):
                r = self._reducer(arenas[0])
                r.begin()
                self.model.grad_ready_hook = r.on_ready
                reducers = [r]
                dp_updater = self._updater_dp(r)
                if dp_updater is not None:
                    self.optimizer.grad_scale = 1.0 / (hvd.size() * accum)
                    dp_updater.begin()
                    r.on_launched = dp_updater.on_bucket
        updater = self._updater() if (last and not self.use_horovod) else None
        if updater is not None:
            self.optimizer.grad_scale = 1.0 / accum
            updater.begin()
            self.model.grad_ready_hook = updater.on_ready
        reserve = bool(reducers) and getattr(self, "reserve_cus_in_backward", True)
        if reserve:
            # the bucketed exchange runs beside the GEMMs from here on: leave RCCL's channel kernels their CUs
            from . import ops
            ops.reserve_cus(True)
        try:
            self.backward_from_loss(accumulate=not first)
        finally:
            if reserve:
                ops.reserve_cus(False)
        self.step_counter_micro = micro + 1
        self._exposed_mark(0)
        if updater is not None:
            self.model.grad_ready_hook = None
            updater.finish()
            return loss_value
        if not last:
            return loss_value

        # gradients hold the SUM over ranks and micro-steps; the mean is taken inside the fused optimizer kernel
        scale = 1.0 / (hvd.size() * accum)
        split_at, owned = None, None
        if self.use_horovod:
            if not reducers:
                reducers = [self._reducer(a) for a in self._arenas()]
                for r in reducers:
                    r.begin()
            if self._dp_mode() == "rs":
                reducers[0].finish()
                owned = reducers[0].owned_ranges()
            elif dp_updater is not None:
                # every bucket's update was queued behind its all-reduce; flush the rest and join
                reducers[0].finish()
                reducers[0].on_launched = None
                if hasattr(self.model, "grad_ready_hook"):
                    self.model.grad_ready_hook = None
                dp_updater.finish()
                self._exposed_mark(1)
                return loss_value
            else:
                # One arena, a fused optimizer and nothing that needs all gradients at once: keep the last
                # bucket (the embeddings) in flight and update everything above it meanwhile.
                can_split = (len(reducers) == 1 and self.post_process_grads is None and
                             isinstance(self.optimizer, Adam) and not self.optimizer.global_clipnorm and
                             os.environ.get("POLUS_DP_SPLIT_ADAM", "1") != "0" and
                             all(v.arena.grads is reducers[0].grads for v in self.trainable_weights))
                split_at = reducers[0].finish(keep_last=True) if can_split else None
                if not can_split:
                    for r in reducers:
                        r.finish()
            if hasattr(self.model, "grad_ready_hook"):
                self.model.grad_ready_hook = None

        grads = [v.grad for v in self.trainable_weights]
        if self.post_process_grads is not None:
            if scale != 1.0:
                from . import ops
                for a in self._arenas():
                    ops.scale_(a.grads, scale)
                scale = 1.0
            grads = self.post_process_grads(grads)
        if hasattr(self.optimizer, "grad_scale"):
            self.optimizer.grad_scale = scale
        if owned is not None:
            # every rank updates its slices of the arena, then the updated f32 parameters travel back
            arena = self._arenas()[0]
            self.optimizer.apply_gradients(list(zip(grads, self.trainable_weights)), _ranges=owned, _refresh=False)
            self._opt_state_synced = False       # this rank's Adam moments are now current on its own slices only
            reducers[0].allgather(arena.params)
            arena.refresh_shadow()
        elif split_at is not None:
            upper = [v for v in self.trainable_weights if v.offset >= split_at]
            lower = [v for v in self.trainable_weights if v.offset < split_at]
            self.optimizer.apply_gradients([(v.grad, v) for v in upper], _refresh=not lower)
            reducers[0].finish_last()
            if lower:
                self.optimizer.apply_gradients([(v.grad, v) for v in lower], _advance=False)
        else:
            self.optimizer.apply_gradients(zip(grads, self.trainable_weights))
        self._exposed_mark(1)
        return loss_value

    def tune_data_parallel(self, one_step, steps=4, bucket_mb=(16, 32, 128)):
        """Measured choice of the two data-parallel switches whose best setting depends on the node at hand -- how RCCL's
        channel kernels and the GEMMs share its CUs, what its links deliver per message size:
          * `reserve_cus_in_backward`: whether the GEMM launches of an exchanging backward pass leave CUs to the channels
            (comm.init / DESIGN.md section 5);
          * `bucket_mb`: the bucket size of the exchange, POLUS_BUCKET_MB (default 64) against the sizes given (an explicit
            POLUS_BUCKET_MB in the environment is kept).
        `one_step()` runs one optimizer step (all its micro-steps) on this rank's next batch; every rank calls this at the same
        point.  (1 + steps) real training steps are taken per candidate; a candidate replaces the incumbent when its slowest
        rank was faster (the times are maxima over the ranks, hence identical everywhere, and so is the choice).  The reserve
        changes tile shapes only -- same results; another bucket size changes where RCCL cuts a message, hence the last
        bits of the summed gradients, as any change of POLUS_BUCKET_MB does.
        Returns the measured times and the choice, or None when there is nothing to choose."""
        import time
        import torch
        if not self.use_horovod or hvd.size() == 1 or not torch.cuda.is_available():
            return None

        def measure():
            one_step()
            torch.cuda.synchronize()
            comm.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            torch.cuda.synchronize()
            return comm.max_over_ranks(time.perf_counter() - t0) / steps * 1e3

        out = {}
        per = {}
        for setting in (True, False):
            self.reserve_cus_in_backward = setting
            per[setting] = measure()
        self.reserve_cus_in_backward = per[True] <= per[False]
        out.update(reserve_on_ms=round(per[True], 3), reserve_off_ms=round(per[False], 3),
                   reserve_cus_in_backward=self.reserve_cus_in_backward)
        if "POLUS_BUCKET_MB" not in os.environ and bucket_mb:
            best_mb, best = 64, min(per.values())
            out["bucket_64_ms"] = round(best, 3)
            for mb in bucket_mb:
                self.bucket_mb = mb
                self._reducers.clear()
                t = measure()
                out[f"bucket_{mb}_ms"] = round(t, 3)
                if t < best:
                    best_mb, best = mb, t
            self.bucket_mb = best_mb
            self._reducers.clear()
            out["bucket_mb"] = best_mb
        return out

    def _exposed_mark(self, which):
        """`trainer.measure_exposed = True` (bench.py, data-parallel runs): HIP events on the compute stream right behind
        the last kernel of backward and behind the last thing the step queues -- the rest of the gradient exchange and
        the optimizer update that nothing overlapped.  `trainer.exposed_events` = (begin, end) of the last step."""
        if not (self.use_horovod and getattr(self, "measure_exposed", False)):
            return
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if which == 0:
            self._exposed_begin = ev
        elif getattr(self, "_exposed_begin", None) is not None:
            self.exposed_events = (self._exposed_begin, ev)
            self._exposed_begin = None

    def lr_finder(self, tf_dataset, use_lr_found=False):
        pass

    def changing_train_config(self, **config):
        for k, v in config.items():
            self.train_config[k] = v

    def sync_optimizer_state(self):
        """Data-parallel runs on the reduce-scatter scheme update, on every rank, only the arena slices that rank
        owns, so a rank's optimizer moments are current only there.  This all-gathers the owned slices of every
        moment: afterwards each rank holds the full, current optimizer state, as it does under Horovod.  Needed
        before anything reads whole moments -- the per-epoch broadcast of polus/training.py:208-211 (rank 0's
        copy would overwrite the other ranks' current slices with stale ones) and save_training_state.
        Collective: every rank must call it.  No-op in a single process, on the all-reduce scheme, or when
        nothing has been updated since the last call."""
        if not self.use_horovod or self._dp_mode() != "rs" or getattr(self, "_opt_state_synced", True):
            return
        arena = self._arenas()[0]
        if hasattr(self.optimizer, "_slots"):
            reducer = self._reducer(arena)
            for moment in self.optimizer._slots(arena):
                reducer.allgather(moment)
        self._opt_state_synced = True

    def broadcast_init_vars(self):
        """polus/training.py:208-211."""
        self.sync_optimizer_state()
        hvd.broadcast_variables(self.trainable_weights, root_rank=0)
        hvd.broadcast_variables(self.optimizer.variables(), root_rank=0)

    # ---- training loop
    _MISSING = {"tf_dataset": "You need to pass a training dataset to the trainer.train method",
                "epochs": "You need to pass the epochs variable to the trainer.train method"}

    def _resolve_train_args(self, given):
        """Arguments of train(): an explicit one wins, then `train_config` (changing_train_config), then -- for the
        dataset and the epoch count only -- the ValueError of polus/training.py:251,257 (text is API)."""
        resolved = {}
        for name, value in given.items():
            unset = value is None or (name == "callbacks" and len(value) == 0)
            if unset and name in self.train_config:
                value = self.train_config[name]
            elif unset and name in self._MISSING:
                raise ValueError(self._MISSING[name])
            resolved[name] = value
        return resolved

    def _one_epoch(self, epoch, dataset, transform):
        """Hook order of polus/training.py:305-330: on_train_batch_begin fires BEFORE the fetch (so once more when the
        iterator is exhausted), the rank-0 broadcast precedes the first step of every epoch, step_counter is global."""
        batches = iter(dataset)
        in_epoch = 0
        while True:
            self.callbacks.on_train_batch_begin(epoch, in_epoch)
            batch = next(batches, None)
            if batch is None:
                return
            if transform is not None:
                batch = transform(batch)
            if in_epoch == 0 and self.use_horovod:
                self.broadcast_init_vars()
            loss = self.train_step(*batch)
            self.callbacks.on_train_batch_end(epoch, in_epoch, loss)
            self.step_counter += 1
            in_epoch += 1
            if self.early_stop:                  # polled after a step, never before the first one
                return

    def train(self, tf_dataset=None, epochs=None, callbacks=[], train_map_f=None, steps=None, **kwargs):
        """polus/training.py:213-338: same signature, same hook order (see _one_epoch), same early-stop polling
        (after a step and after an epoch's on_epoch_end)."""
        a = self._resolve_train_args(dict(tf_dataset=tf_dataset, epochs=epochs, callbacks=callbacks,
                                          train_map_f=train_map_f, steps=steps))
        dataset, n_epochs, hooks, transform = a["tf_dataset"], a["epochs"], a["callbacks"], a["train_map_f"]
        transform = kwargs.pop("custom_data_transform_f", transform)          # legacy spelling (polus/training.py:275-276)

        steps_per_epoch = a["steps"]
        if steps_per_epoch is None:
            try:
                steps_per_epoch = len(dataset)
            except TypeError:
                steps_per_epoch = -2                                          # tf.data UNKNOWN_CARDINALITY

        if os.getenv("POLUS_PROFILER", "False").lower() in ("true", "1", "t", "y", "yes"):
            lo_hi = [int(x) for x in os.getenv("POLUS_PROFILER_RANGE", "10:20").split(":")]
            logger.info("POLUS_PROFILER env was set to True, so the Profiler callback was added to training")
            hooks = list(hooks) + [Profiler(steps_interval=lo_hi)]            # a copy: the caller's list is left alone

        if not isinstance(hooks, CallbackCoordinator):
            hooks = CallbackCoordinator(hooks, trainer=self, epochs=n_epochs, steps=steps_per_epoch)
        self.callbacks = hooks

        self.callbacks.on_train_begin()
        for epoch in range(n_epochs):
            self.callbacks.on_epoch_begin(epoch)
            self._one_epoch(epoch, dataset, transform)
            self.callbacks.on_epoch_end(epoch)
            if self.early_stop:
                break
        self.callbacks.on_train_end()


class ClassifierTrainer(BaseTrainer):
    """polus/training.py:341-397."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def forward_with_grads(self, x, y):
        if isinstance(x, dict):
            logits = self.model(**x, training=True)
        else:
            logits = self.model(x, training=True)
        if self.post_process_logits is not None:
            logits = self.post_process_logits(logits)
        return y, logits


Trainer = ClassifierTrainer  # BASELINE.json's wording; the reference has no class of this name
