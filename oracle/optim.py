"""NumPy restatement of the optimizer / schedule step (test infrastructure only).

  * polus/training.py:191        optimizer.apply_gradients(zip(grads, weights))
  * polus/schedulers.py:5-23     WarmUp(linear) -> PolynomialDecay(power=1, end_lr=1e-7
                                 hard-coded; the ``end_lr`` argument is ignored)
  * Keras ``Adam`` (tutorials/classifier_example.py:54) and HF ``AdamWeightDecay``
    (imported at polus/schedulers.py:2) are third-party and absent here: restated
    from their published update rules.  Parity unpinned.

Keras Adam (non-amsgrad), t = iterations + 1:
    lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
    p = p - lr_t * m / (sqrt(v) + eps)            eps = 1e-7, outside the bias correction
HF AdamWeightDecay: before that update, for every variable whose name does not
match an exclusion pattern (LayerNorm / layer_norm / bias):  p = p - lr * wd * p
(lr = the scheduled, not bias-corrected, rate).
"""
import math

import numpy as np


def warmup_linear_lr(step, num_train_steps, max_lr, warmup_percentage=0.1):
    """polus/schedulers.py:10-23 evaluated at integer `step` (the optimizer's iterations)."""
    warm = int(num_train_steps * warmup_percentage)
    end = 1e-7
    if step < warm:
        return max_lr * (step / warm)
    decay_steps = num_train_steps - warm
    s = min(step - warm, decay_steps)
    return (max_lr - end) * (1.0 - s / decay_steps) + end


class Adam:
    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, weight_decay=0.0,
                 no_decay=()):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.no_decay = set(no_decay)
        self.t = 0
        self.m, self.v = {}, {}

    def lr_at(self, t):
        return self.lr(t) if callable(self.lr) else self.lr

    def step(self, params, grads):
        lr = self.lr_at(self.t)  # schedule is evaluated at `iterations` before increment
        self.t += 1
        t = self.t
        lr_t = lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        for k, g in grads.items():
            p = params[k]
            dt = p.dtype.type
            if k not in self.m:
                self.m[k] = np.zeros_like(p)
                self.v[k] = np.zeros_like(p)
            if self.wd and k not in self.no_decay:
                p -= dt(lr * self.wd) * p
            m, v = self.m[k], self.v[k]
            m *= dt(self.b1); m += dt(1.0 - self.b1) * g
            v *= dt(self.b2); v += dt(1.0 - self.b2) * g * g
            p -= dt(lr_t) * m / (np.sqrt(v) + dt(self.eps))


def is_no_decay(name):
    """HF AdamWeightDecay exclude_from_weight_decay=["LayerNorm","layer_norm","bias"],
    mapped onto this repo's parameter names (ln*.g / ln*.b / *.b)."""
    return name.endswith(".b") or ".ln" in name


def clip_by_global_norm(grads, clip_norm):
    gn = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    scale = clip_norm / max(gn, clip_norm)
    return {k: g * g.dtype.type(scale) for k, g in grads.items()}, gn


# ----------------------------------------------------------------------------- metrics / DP rules

def confusion_matrix(y_true, y_pred, num_classes):
    """polus/metrics.py:51-59 — rows = first argument."""
    cm = np.zeros((num_classes, num_classes), np.int32)
    np.add.at(cm, (np.asarray(y_true).reshape(-1), np.asarray(y_pred).reshape(-1)), 1)
    return cm


def macro_f1(cm):
    """polus/metrics.py:76-90 (float64, divide_no_nan)."""
    def dnn(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        out = np.zeros(np.broadcast(a, b).shape)
        np.divide(a, b, out=out, where=(b != 0))
        return out
    tp = np.diag(cm).astype(np.float64)
    fp_tp = cm.sum(-1).astype(np.float64)
    fn_tp = cm.sum(-2).astype(np.float64)
    precision, recall = dnn(tp, fp_tp), dnn(tp, fn_tp)
    inv_p, inv_r = dnn(1.0, precision), dnn(1.0, recall)
    return float(np.mean(dnn(2.0, inv_p + inv_r)))


def shard_indices(n_samples, world, rank):
    """polus/data.py:94-96 — Dataset.shard(num_shards=size, index=local_rank):
    element i goes to rank i mod size, before batching."""
    return list(range(rank, n_samples, world))
