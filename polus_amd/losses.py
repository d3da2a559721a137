"""Loss objects of the Polus step: ``loss(*forward_with_grads_outputs) -> scalar``
(polus/training.py:180) plus ``backward()`` -> gradient wrt the logits, both one HIP launch.

polus/losses.py:5-41 and the Keras SparseCategoricalCrossentropy(from_logits=True) of
tutorials/classifier_example.py:55.
"""
import numpy as np
import torch

from . import ops
from .tensor import DeviceScalar, to_device


class _XentBase:
    RING = 256

    def __init__(self, grad_dtype=None):
        self.grad_dtype = grad_dtype
        self._bufs = {}

    def _out(self, logits):
        l2 = logits.reshape(-1, logits.shape[-1])
        if l2.dtype != torch.float32:
            l2 = l2.float()
        l2 = l2.contiguous()
        gd = self.grad_dtype or torch.float32
        key = (tuple(l2.shape), gd)
        if self._bufs.get("key") != key:
            self._bufs = {"key": key, "d": torch.empty(l2.shape, dtype=gd, device=l2.device),
                          "ring": torch.empty(self.RING, dtype=torch.float32, device=l2.device), "k": 0}
        # every call hands out its own scalar: a loss kept from step k still reads step k's value after later
        # steps (the reference returns independent tf tensors); the ring wraps after RING steps
        k = self._bufs["k"]
        self._bufs["k"] = (k + 1) % self.RING
        return l2, self._bufs["d"], self._bufs["ring"][k:k + 1]

    def backward(self, accumulate=False):
        return self._dlogits


class SparseCategoricalCrossentropy(_XentBase):
    """mean over every leading dim of -log softmax(logits)[label]."""

    def __init__(self, from_logits=True, grad_dtype=None):
        assert from_logits, "only from_logits=True is on the reference's path"
        super().__init__(grad_dtype)

    def __call__(self, y_true, y_pred):
        l2, d, loss = self._out(y_pred)
        labels = to_device(y_true, torch.int32, l2.device).reshape(-1)
        ops.softmax_xent(l2, labels, loss, d)
        self._dlogits = d.view(y_pred.shape)
        return DeviceScalar(loss)


def weighted_softmax_cross_entropy_from_logits(class_weights, grad_dtype=None):
    """polus/losses.py:5-18; y_true one-hot."""
    cw_host = np.asarray(class_weights, np.float32)

    class _Loss(_XentBase):
        def __call__(self, y_true, y_pred):
            l2, d, loss = self._out(y_pred)
            yt = to_device(y_true, None, l2.device)
            labels = yt.reshape(-1, yt.shape[-1]).argmax(-1).to(torch.int32).contiguous()
            cw = to_device(cw_host, torch.float32, l2.device)
            ops.softmax_xent(l2, labels, loss, d, class_weights=cw)
            self._dlogits = d.view(y_pred.shape)
            return DeviceScalar(loss)
    return _Loss(grad_dtype)


def weighted_sigmoid_cross_entropy_from_logits(class_weights, negative_weight, grad_dtype=None):
    """polus/losses.py:21-41; y_true multi-hot."""
    cw_host = np.asarray(class_weights, np.float32)

    class _Loss(_XentBase):
        def __call__(self, y_true, y_pred):
            l2, d, loss = self._out(y_pred)
            yt = to_device(y_true, torch.float32, l2.device).reshape(l2.shape)
            cw = to_device(cw_host, torch.float32, l2.device)
            ops.sigmoid_xent(l2, yt, cw, negative_weight, loss, d)
            self._dlogits = d.view(y_pred.shape)
            return DeviceScalar(loss)
    return _Loss(grad_dtype)
