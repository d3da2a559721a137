"""CPU test doubles for the host-logic tests: a tiny NumPy softmax-regression 'model' that
implements the model / loss / optimizer protocols the trainer drives (the oracle is the
compute here — test infrastructure only)."""
import numpy as np
import torch

from oracle import losses as ol
from polus_amd.optimizers import _LearningRate


class FakeArena:
    def __init__(self, n):
        self.grads = torch.zeros(n, dtype=torch.float32)
        self.params = torch.zeros(n, dtype=torch.float32)
        self.shadow = None
        self.vars = []

    def refresh_shadow(self, var=None):
        pass


class FakeVar:
    def __init__(self, arena, name, offset, shape):
        self.arena, self.name, self.offset, self.shape = arena, name, offset, shape
        self.size = int(np.prod(shape))

    @property
    def value(self):
        return self.arena.params[self.offset:self.offset + self.size].view(self.shape)

    @property
    def grad(self):
        return self.arena.grads[self.offset:self.offset + self.size].view(self.shape)


class FakeLinearModel:
    """logits = x W^T + b with NumPy math; grads land in a flat CPU 'arena'."""
    name = "fake"

    def __init__(self, n_in=5, n_out=3, seed=0):
        self.arena = FakeArena(n_in * n_out + n_out)
        self.w = FakeVar(self.arena, "w", 0, (n_out, n_in))
        self.b = FakeVar(self.arena, "b", n_in * n_out, (n_out,))
        self.arena.vars = [self.w, self.b]
        self.arena.params[:n_in * n_out] = torch.from_numpy(
            np.random.default_rng(seed).standard_normal(n_in * n_out).astype(np.float32))
        self.trainable_weights = [self.w, self.b]
        self.grad_ready_hook = None
        self.calls = []

    def __call__(self, x, training=False):
        self.calls.append(("call", training))
        self._x = np.asarray(x, np.float64)
        return self._x @ self.w.value.numpy().astype(np.float64).T + self.b.value.numpy()

    def backward(self, dlogits, accumulate=False):
        gw = torch.from_numpy((dlogits.T @ self._x).astype(np.float32))
        gb = torch.from_numpy(dlogits.sum(0).astype(np.float32))
        if accumulate:
            self.w.grad.add_(gw); self.b.grad.add_(gb)
        else:
            self.w.grad.copy_(gw); self.b.grad.copy_(gb)
        if self.grad_ready_hook:
            self.grad_ready_hook(self.b.offset, self.b.offset + self.b.size)
            self.grad_ready_hook(self.w.offset, self.w.offset + self.w.size)

    def inference(self, x):
        return np.argmax(self(x), -1).astype(np.int32)

    def save(self, **kw):
        self.calls.append(("save", kw))


class FakeXent:
    def __call__(self, y, logits):
        self.loss, self.d = ol.sparse_softmax_xent_fwd(np.asarray(logits, np.float64), np.asarray(y))
        return float(self.loss)

    def backward(self, accumulate=False):
        return self.d


class FakeSGD:
    def __init__(self, lr=0.1):
        self.learning_rate = _LearningRate(lr)
        self.grad_scale = 1.0
        self.steps = 0

    def variables(self):
        return []

    def apply_gradients(self, grads_and_vars):
        lr = self.learning_rate(self.steps)
        self.steps += 1
        for g, v in grads_and_vars:
            v.value.sub_(lr * self.grad_scale * g)
