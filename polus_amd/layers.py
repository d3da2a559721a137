"""Layers with explicit forward/backward (no autograd, no tape): every backward is a fixed
sequence of HIP launches, which is what lets the gradient all-reduce start the moment a
layer's last dW kernel is queued.

Mirrors the Keras layers the reference composes (tutorials/classifier_example.py:44-48,
polus/ner/models.py:35-39) and polus/layers.py CRF.
"""
import math
import os
import zlib

import numpy as np
import torch

from . import ops
from .tensor import DeviceScalar, to_device


class Layer:
    """forward(x, training) -> y ; backward(dy) -> dx ; variables() -> [Variable]."""
    name = "layer"

    def build(self, arena, in_features, prefix):
        return in_features

    def variables(self):
        return []

    def forward(self, x, training=False):
        raise NotImplementedError

    def backward(self, dy, accumulate=False):
        raise NotImplementedError

    def _buf(self, key, shape, dtype, dev):
        cache = self.__dict__.setdefault("_bufs", {})
        t = cache.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype:
            t = cache[key] = torch.empty(tuple(shape), dtype=dtype, device=dev)
        return t


def dw_split_k(out_rows, out_cols, contraction):
    """Split-K factor for dW = dY^T X: few 256x128 output tiles, long contraction (T).  Aim at
    one full round of the 512 workgroup slots (2 per CU) while keeping >= 256 contraction rows
    per slice."""
    tiles = ((out_rows + 255) // 256) * ((out_cols + 127) // 128)
    want = max(1, 512 // max(tiles, 1))
    return int(max(1, min(want, contraction // 256, 64)))


def dw_group_split_k(shapes, contraction):
    """Split-K factor for one grouped dW launch over `shapes` = [(n_out, n_in), ...]: the
    concatenated 256x128 tile lists (each padded to a multiple of 8) times the splits should
    fill the 512 workgroup slots once."""
    tiles = sum((((o + 255) // 256) * ((i + 127) // 128) + 7) // 8 * 8 for o, i in shapes)
    want = max(1, 512 // max(tiles, 1))
    return int(max(1, min(want, contraction // 256, 64)))


def dense_bwd_params_group(problems, accumulate=False):
    """dW (+ db) of several Dense layers that share T: one grouped launch on the bf16 engine; the
    f32 engine (128x128 exact-f32 kernel) runs them one by one with per-matrix split-K."""
    T = problems[0][0].shape[0]
    if problems[0][0].dtype == torch.bfloat16 and os.environ.get("POLUS_DW_UNGROUPED") is None:
        return ops.dense_bwd_params_grouped(problems, accumulate, 0)      # 0: per-problem splits chosen by the library
    for dy, x, dw, db in problems:
        sk = dw_split_k(dy.shape[1], x.shape[1], T)
        if db is not None:
            ops.dense_bwd_params(dy, x, dw, db, accumulate, sk)
        else:
            ops.gemm(dy, x, dw, a_layout=ops.K_STRIDED, b_layout=ops.K_STRIDED,
                     flags=ops.GEMM_ACCUM_C if accumulate else 0, split_k=sk)


def gemm_dx(dy, w, dx, **kw):
    """dX = dY . W for W stored [out, in].  bf16 engine: the transposed shadow W^T [in, out]
    (refreshed once per optimizer step) is a K-contiguous B operand; otherwise W itself is the
    K-strided operand (k-major LDS image, transposing LDS reads)."""
    wt = w.compute_t
    if wt is not None:
        return ops.gemm(dy, wt, dx, split_k="auto", **kw)
    return ops.gemm(dy, w.compute, dx, b_layout=ops.K_STRIDED, **kw)


def glorot_uniform(rng, fan_in, fan_out, shape):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


class Flatten(Layer):
    """tf.keras.layers.Flatten: [B, ...] -> [B, prod]."""

    def __init__(self, input_shape=None):
        self.input_shape = input_shape

    def build(self, arena, in_features, prefix):
        if self.input_shape is not None:
            return int(np.prod(self.input_shape))
        return in_features

    def forward(self, x, training=False):
        self._shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, dy, accumulate=False):
        return dy.reshape(self._shape)


class Dropout(Layer):
    """tf.keras.layers.Dropout: inverted dropout in training, identity otherwise.  The mask is a
    hash of (seed, element index) regenerated in backward; the seed advances every training call."""
    _instances = 0

    def __init__(self, rate=0.0, input_shape=None, seed=None):
        self.rate = float(rate)
        if not 0.0 <= self.rate < 1.0:
            raise ValueError("dropout rate must be in [0, 1)")
        Dropout._instances += 1
        self.seed = (seed if seed is not None else 7919 * Dropout._instances) & 0xFFFFFFFF
        self.calls = 0
        self._active = None

    def forward(self, x, training=False):
        if not training or self.rate == 0.0:
            self._active = None
            return x
        self.calls += 1
        s = (self.seed + 0x9E3779B1 * self.calls) & 0xFFFFFFFF
        xc = x.contiguous()
        y = self._buf("y", xc.shape, xc.dtype, xc.device)
        ops.dropout(xc, y, self.rate, s)
        self._active = s
        return y

    def backward(self, dy, accumulate=False):
        if self._active is None:
            return dy
        dyc = dy.contiguous()
        dx = self._buf("dx", dyc.shape, dyc.dtype, dyc.device)
        ops.dropout(dyc, dx, self.rate, self._active)
        return dx


class Dense(Layer):
    """Keras Dense: y = act(x W^T + b), W stored [out, in].  out_dtype=float32 makes the
    layer emit f32 (logits feeding a loss)."""

    def __init__(self, units, activation=None, input_shape=None, use_bias=True, out_dtype=None, name=None):
        self.units = int(units)
        self.activation = activation if activation not in ("linear",) else None
        self.input_shape = input_shape
        self.use_bias = use_bias
        self.out_dtype = out_dtype
        self.name = name or "dense"

    def build(self, arena, in_features, prefix):
        if in_features is None:
            in_features = self.input_shape[-1]
        self.in_features = int(in_features)
        rng = np.random.Generator(np.random.PCG64(zlib.crc32(f"{prefix}:{self.units}:{self.in_features}".encode())))
        self.w = arena.add(prefix + ".w", (self.units, self.in_features),
                           glorot_uniform(rng, self.in_features, self.units, (self.units, self.in_features)),
                           decay=True, matrix=True)
        self.b = arena.add(prefix + ".b", (self.units,), np.zeros(self.units, np.float32), decay=False) if self.use_bias else None
        self.arena = arena
        return self.units

    def variables(self):
        return [self.w] + ([self.b] if self.b is not None else [])

    def forward(self, x, training=False):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        assert x2.shape[1] == self.in_features, f"{self.name}: expected {self.in_features} features, got {x2.shape[1]}"
        rows = x2.shape[0]
        odt = self.out_dtype or x2.dtype
        y = self._buf("y", (rows, self.units), odt, x2.device)
        bias = self.b.value if self.b is not None else None
        # a handful of units (a classification head): HBM-bound wave-per-row kernels, no 128-wide MFMA tiles of padding
        self._thin = (not self.activation and x2.is_cuda and x2.stride(1) == 1 and
                      (x2.data_ptr() % 16 == 0) and ((x2.stride(0) * x2.element_size()) % 16 == 0) and
                      os.environ.get("POLUS_DENSE_THIN", "1") != "0" and ops.dense_thin_supported(x2.dtype, self.in_features, self.units))
        if self._thin:
            ops.dense_thin_fwd(x2, self.w.compute, bias, y)
            self._x = x2
            return y.view(*lead, self.units)
        if self.activation:
            assert odt == x2.dtype, "an activated Dense keeps the compute dtype"
            u = self._buf("u", (rows, self.units), x2.dtype, x2.device)
            ops.gemm(x2, self.w.compute, y, bias=bias, aux=u, act=self.activation, flags=ops.GEMM_ACT_FWD, split_k="auto")
            self._u = u
        else:
            ops.gemm(x2, self.w.compute, y, bias=bias, split_k="auto")
        self._x = x2
        return y.view(*lead, self.units)

    def backward(self, dy, accumulate=False, need_dx=True, dx_resid=None):
        """dx_resid [rows, in] (row stride free): added to dX in the GEMM epilogue."""
        x = self._x
        dy2 = dy.reshape(-1, self.units)
        if getattr(self, "_thin", False) and dx_resid is None and dy2.dtype in (x.dtype, torch.float32):
            rows = x.shape[0]
            dx = self._buf("dx", (rows, self.in_features), x.dtype, x.device) if need_dx else None
            ops.dense_thin_bwd(x, dy2 if dy2.stride(1) == 1 else dy2.contiguous(), self.w.compute, dx, self.w.grad,
                               self.b.grad if self.b is not None else None, accumulate)
            return dx
        if dy2.dtype != x.dtype:  # f32 dlogits of an f32-output layer in bf16 mode
            d = self._buf("dy_cast", dy2.shape, x.dtype, x.device)
            ops.cast(dy2.contiguous(), d)
            dy2 = d
        if self.activation:
            du = self._buf("du", dy2.shape, x.dtype, x.device)
            ops.act_bwd(dy2.contiguous(), self._u, du, self.activation)
            dy2 = du
        rows = x.shape[0]
        acc = ops.GEMM_ACCUM_C if accumulate else 0
        ops.dense_bwd_params(dy2.contiguous() if dy2.stride(1) != 1 else dy2, x, self.w.grad,
                             self.b.grad if self.b is not None else None, accumulate,
                             dw_split_k(self.units, self.in_features, rows))
        if not need_dx:
            return None
        dx = self._buf("dx", (rows, self.in_features), x.dtype, x.device)
        if dx_resid is not None:
            gemm_dx(dy2, self.w, dx, resid=dx_resid)
        else:
            gemm_dx(dy2, self.w, dx)
        return dx


class CRF(Layer):
    """polus/layers.py:6-140.  Training: passes the potentials through and remembers the
    sequence lengths (full S when none are given, :74-76); inference: Viterbi decode to
    one-hot (:78-84).  `loss` / `loss_sample_weights` build the NLL of :86-126."""

    def __init__(self, output_dim, sparse_target=True, mask_impossible_transitions=None, name="crf"):
        self.output_dim = int(output_dim)
        self.mask_impossible_transitions = mask_impossible_transitions
        self.sequence_lengths = None
        self.name = name

    def build(self, arena, in_features, prefix):
        if in_features != self.output_dim:
            raise ValueError("The last dimension of the input shape must be equal to output shape. "
                             "Use a linear layer if needed.")
        C = self.output_dim
        rng = np.random.Generator(np.random.PCG64(77 + C))
        self.transitions = arena.add(prefix + ".transitions", (C, C), glorot_uniform(rng, C, C, (C, C)), decay=True)
        self.arena = arena
        return C

    def variables(self):
        return [self.transitions]

    def get_transitions(self):
        """polus/layers.py:58-63 — T*M + (1-M)*-10000; a tiny [C,C] host-side parameter
        transform re-uploaded per step (C <= 16)."""
        t = self.transitions.value
        if self.mask_impossible_transitions is None:
            return t
        m = to_device(np.asarray(self.mask_impossible_transitions, np.float32), torch.float32, t.device)
        return t * m + (1.0 - m) * -10000.0

    def forward(self, x, training=False, sequence_lengths=None):
        assert x.dim() == 3 and x.shape[-1] == self.output_dim
        B, S, C = x.shape
        if sequence_lengths is not None:
            sl = to_device(sequence_lengths, torch.int32, x.device).reshape(-1)
        else:
            sl = torch.full((B,), S, dtype=torch.int32, device=x.device)
        self.sequence_lengths = sl
        self._pot = x
        if training:
            return x
        tags = self._buf("tags", (B, S), torch.int32, x.device)
        ops.crf_viterbi(x.float().contiguous() if x.dtype != torch.float32 else x.contiguous(), sl,
                        self.get_transitions().contiguous(), tags)
        return torch.nn.functional.one_hot(tags.long(), C).to(torch.float32)

    def backward(self, dy, accumulate=False):
        return dy

    def _nll(self, y_true, y_pred, sample_weights):
        pot = y_pred if y_pred.dtype == torch.float32 else y_pred.float()
        pot = pot.contiguous()
        B, S, C = pot.shape
        yt = to_device(y_true, None, pot.device)
        tags = (yt.argmax(-1) if yt.dim() == 3 else yt).to(torch.int32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pot.device)
        dpot = self._buf("dpot", (B, S, C), torch.float32, pot.device)
        dtr = self._buf("dtrans", (C, C), torch.float32, pot.device)
        ops.crf_nll(pot, tags, self.sequence_lengths, self.get_transitions().contiguous(), sample_weights, loss, dpot, dtr)
        if self.mask_impossible_transitions is not None:
            dtr = dtr * to_device(np.asarray(self.mask_impossible_transitions, np.float32), torch.float32, pot.device)
        self._dtrans, self._dpot = dtr, dpot
        return DeviceScalar(loss)

    @property
    def loss(self):
        layer = self

        class _CRFLoss:
            def __call__(self, y_true, y_pred):
                return layer._nll(y_true, y_pred, None)

            def backward(self, accumulate=False):
                g = layer.transitions.grad
                g.add_(layer._dtrans) if accumulate else g.copy_(layer._dtrans)
                return layer._dpot
        return _CRFLoss()

    def loss_sample_weights(self, mask_positive_classes, negative_weight):
        """polus/layers.py:101-126: per-sequence weight 1 if any positive class occurs else
        negative_weight (computed on the labels, host side of the step)."""
        layer = self
        mpc = np.asarray(mask_positive_classes, np.float32)

        class _CRFWLoss:
            def __call__(self, y_true, y_pred):
                yt = to_device(y_true, torch.float32, y_pred.device)
                pos = yt * to_device(mpc, torch.float32, yt.device)
                neg = (pos == 0).all(-1).all(-1)
                w = (pos == 1).any(-1).any(-1).to(torch.float32) + neg.to(torch.float32) * float(negative_weight)
                return layer._nll(y_true, y_pred, w.contiguous())

            def backward(self, accumulate=False):
                g = layer.transitions.grad
                g.add_(layer._dtrans) if accumulate else g.copy_(layer._dtrans)
                return layer._dpot
        return _CRFWLoss()
