"""Optimizers and learning-rate schedules of the Polus step, fused over the flat arena.

Replaces Keras ``Adam`` (tutorials/classifier_example.py:54) and HF ``AdamWeightDecay``
(imported at polus/schedulers.py:2) behind the protocol ``BaseTrainer`` uses
(polus/training.py:90-94, :191, :211): ``learning_rate.read_value()/assign()``,
``apply_gradients(zip(grads, vars))``, ``variables()``.
One HIP launch updates every tensor of an arena (28 B/param of HBM traffic, + 2 B for the
bf16 shadow of GEMM weights).
"""
import math

import numpy as np
import torch

from . import ops

_CHUNK = 1 << 14


class _LearningRate:
    """Keras-variable-like handle (polus/training.py:91-93)."""

    def __init__(self, value):
        self._value = value

    def read_value(self):
        if callable(self._value):
            raise TypeError("learning rate is a schedule; scale its max_lr instead")
        return self._value

    def numpy(self):
        return self.read_value()

    def assign(self, v):
        self._value = float(v)

    def __call__(self, step):
        return float(self._value(step)) if callable(self._value) else float(self._value)

    def scale(self, k):
        """LR x world-size rule of polus/training.py:90-94, also for schedules."""
        if callable(self._value):
            inner = self._value
            self._value = lambda step, inner=inner, k=k: inner(step) * k
        else:
            self._value = self._value * k


class WarmUpLinearDecay:
    """polus/schedulers.py:5-23: linear warm-up over int(N*pct) steps, then linear decay to
    1e-7 (the reference ignores its ``end_lr`` argument and hard-codes 1e-7)."""

    def __init__(self, num_train_steps, max_lr, warmup_percentage=0.1, end_lr=1e-7):
        self.num_train_steps, self.max_lr = num_train_steps, max_lr
        self.num_warmup_steps = int(num_train_steps * warmup_percentage)
        self.decay_steps = num_train_steps - self.num_warmup_steps
        self.end_lr = 1e-7

    def __call__(self, step):
        if step < self.num_warmup_steps:
            return self.max_lr * (step / self.num_warmup_steps)
        s = min(step - self.num_warmup_steps, self.decay_steps)
        return (self.max_lr - self.end_lr) * (1.0 - s / self.decay_steps) + self.end_lr


def default_no_decay(name):
    """HF AdamWeightDecay exclude_from_weight_decay = ["LayerNorm", "layer_norm", "bias"]
    mapped onto this repo's names (ln*.g / ln*.b / *.b)."""
    return name.endswith(".b") or ".ln" in name


class Adam:
    """Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps), eps=1e-7."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, weight_decay_rate=0.0,
                 exclude_from_weight_decay=default_no_decay, global_clipnorm=None, name="Adam"):
        self.learning_rate = _LearningRate(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.weight_decay_rate = weight_decay_rate
        self.exclude = exclude_from_weight_decay
        self.global_clipnorm = global_clipnorm
        self.iterations = 0
        self.grad_scale = 1.0        # set by the trainer: 1/world (SUM all-reduce) and 1/accum
        self._state = {}             # arena id -> (m, v)
        self._tables = {}            # (arena id, var ids) -> segment table
        self.name = name

    # -- protocol used by BaseTrainer / callbacks
    def variables(self):
        out = []
        for m, v in self._state.values():
            out += [m, v]
        return out

    def get_config(self):
        return {"name": self.name, "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "weight_decay_rate": self.weight_decay_rate}

    def _slots(self, arena):
        st = self._state.get(id(arena))
        if st is None:
            st = (torch.zeros_like(arena.params), torch.zeros_like(arena.params))
            self._state[id(arena)] = st
        return st

    def _table(self, arena, variables, ranges=None):
        """Device segment table (begin, end, flags) of the windows to update: the variables, cut into
        chunks, and -- data-parallel sharded update -- intersected with the `ranges` this rank owns."""
        key = (id(arena), tuple(id(v) for v in variables), tuple(ranges) if ranges is not None else None)
        t = self._tables.get(key)
        if t is None:
            seg = []
            for v in variables:
                decay = v.decay and not (self.exclude and self.exclude(v.name))
                flags = (1 if (decay and self.weight_decay_rate) else 0) | (2 if v.matrix else 0)
                windows = [(v.offset, v.offset + v.size)]
                if ranges is not None:
                    windows = [(max(v.offset, lo), min(v.offset + v.size, hi)) for lo, hi in ranges]
                    windows = [(lo, hi) for lo, hi in windows if hi > lo]
                for wlo, whi in windows:
                    for b in range(wlo, whi, _CHUNK):
                        seg.append((b, min(whi, b + _CHUNK), flags))
            if not seg:
                seg = [(0, 0, 0)]
            t = (torch.from_numpy(np.asarray(seg, np.int64)).to(arena.device), len(seg))
            self._tables[key] = t
        return t

    def apply_gradients(self, grads_and_vars, _advance=True, _refresh=True, _ranges=None):
        """`grads` are the arena gradient windows of the variables (the trainer passes
        ``v.grad``); one fused launch per arena.  `_advance=False` applies the SAME step (count,
        learning rate) to further variables -- the data-parallel trainer updates the encoder while
        the embedding gradients are still being all-reduced; `_refresh=False` postpones the
        transposed-shadow refresh to that second call; `_ranges` = the arena windows this rank owns after
        a gradient reduce-scatter (only those are updated; the parameters are all-gathered afterwards)."""
        variables = [v for _, v in grads_and_vars]
        if _advance:
            self._lr_now = self.learning_rate(self.iterations)
            self.iterations += 1
        lr = self._lr_now
        t = self.iterations
        lr_t = lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        arenas = {}
        for v in variables:
            arenas.setdefault(id(v.arena), (v.arena, []))[1].append(v)
        clip = None
        if self.global_clipnorm:
            # tf.clip_by_global_norm: ONE norm over the gradients being applied -- every arena, and only the
            # windows of the variables in this call (frozen or stale gradient windows do not count)
            dev0 = next(iter(arenas.values()))[0].device
            sq = torch.empty(len(arenas), dtype=torch.float32, device=dev0)
            clip = torch.empty(1, dtype=torch.float32, device=dev0)
            for k, (arena, vs) in enumerate(arenas.values()):
                seg, n_seg = self._table(arena, vs, _ranges)
                ops.sqnorm_segments(arena.grads, seg, n_seg, sq[k:k + 1])
            ops.clip_scale(sq, self.grad_scale, self.global_clipnorm, clip)
        for arena, vs in arenas.values():
            m, v = self._slots(arena)
            seg, n_seg = self._table(arena, vs, _ranges)
            ops.adam_step(arena.params, arena.grads, m, v, arena.shadow, seg, n_seg, lr, lr_t,
                          self.beta_1, self.beta_2, self.epsilon, self.weight_decay_rate,
                          grad_scale=self.grad_scale, clip_scale=clip)
            if _refresh:
                arena.refresh_transposed()


class AdamWeightDecay(Adam):
    """HF AdamWeightDecay: decoupled decay p -= lr*wd*p on everything but LayerNorm/bias."""

    def __init__(self, learning_rate=0.001, weight_decay_rate=0.01, **kw):
        super().__init__(learning_rate=learning_rate, weight_decay_rate=weight_decay_rate,
                         name=kw.pop("name", "AdamWeightDecay"), **kw)
