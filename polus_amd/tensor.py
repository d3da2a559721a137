"""Device-memory plumbing: the flat parameter arena, variables, lazy scalars.

torch owns the HBM allocations and nothing else; all arithmetic on them is done by the
HIP kernels of libpolus_hip.so.
"""
import os

import numpy as np
import torch

ALIGN = 64  # elements: every tensor in an arena starts on a 256-byte boundary


def device():
    """The GPU this process drives (one process per GPU, polus/__init__.py:102-127)."""
    if not torch.cuda.is_available():
        raise RuntimeError("polus_amd needs an MI355X: no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device(x, dtype=None, dev=None):
    """numpy / python / torch (any device) -> contiguous device tensor."""
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != dev:
        t = t.to(dev, non_blocking=True)
    return t.contiguous()


class DeviceScalar:
    """A one-element device tensor that only synchronises when a host value is needed
    (callbacks format the loss with ``f"{loss:.3f}"``, polus/callbacks.py:603)."""

    def __init__(self, t):
        self.t = t

    def item(self):
        return float(self.t.item())

    def numpy(self):
        return np.float32(self.item())

    __float__ = item

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return f"DeviceScalar({self.item():.6g})"

    def __lt__(self, o): return self.item() < float(o)
    def __le__(self, o): return self.item() <= float(o)
    def __gt__(self, o): return self.item() > float(o)
    def __ge__(self, o): return self.item() >= float(o)
    def __add__(self, o): return self.item() + float(o)
    __radd__ = __add__
    def __sub__(self, o): return self.item() - float(o)
    def __rsub__(self, o): return float(o) - self.item()
    def __mul__(self, o): return self.item() * float(o)
    __rmul__ = __mul__
    def __truediv__(self, o): return self.item() / float(o)


class Variable:
    """One trainable tensor: a named window of a ParamArena (value, grad, bf16 shadow)."""

    def __init__(self, arena, name, shape, decay, matrix, init):
        self.arena, self.name, self.shape = arena, name, tuple(int(s) for s in shape)
        self.decay, self.matrix, self._init = decay, matrix, init
        self.size = int(np.prod(self.shape))
        self.offset = None

    def _view(self, flat, key, shape=None):
        # views of the flat arenas are made once per (variable, arena tensor): the layers ask for
        # them ~500 times per step
        c = self.__dict__.setdefault("_views", {})
        v = c.get(key)
        if v is None or v[0] is not flat:
            v = c[key] = (flat, flat[self.offset:self.offset + self.size].view(shape or self.shape))
        return v[1]

    @property
    def value(self):
        return self._view(self.arena.params, "p")

    @property
    def grad(self):
        return self._view(self.arena.grads, "g")

    @property
    def compute_t(self):
        """bf16 transposed shadow [in, out] (K-contiguous B operand of dX = dY . W), or None."""
        if self.arena.shadow_t is not None and self.matrix and len(self.shape) == 2:
            return self._view(self.arena.shadow_t, "t", (self.shape[1], self.shape[0]))
        return None

    @property
    def compute(self):
        """The tensor GEMMs read: the bf16 shadow in bf16 mode, the f32 master otherwise."""
        if self.arena.shadow is not None and self.matrix:
            return self._view(self.arena.shadow, "s")
        return self.value

    def numpy(self):
        return self.value.detach().cpu().numpy()

    def assign(self, array):
        self.value.copy_(to_device(array, torch.float32, self.arena.device).view(self.shape))
        self.arena.refresh_shadow(self)

    def __repr__(self):
        return f"Variable({self.name}, shape={self.shape})"


class ParamArena:
    """Flat f32 parameter / gradient storage (+ bf16 shadow of the GEMM weights).

    One arena per model: the optimizer sweeps it with one fused kernel, the gradient
    all-reduce walks it in contiguous buckets (reverse of forward order), and a broadcast of
    the initial weights is one collective."""

    def __init__(self, compute_dtype=torch.float32, dev=None):
        self.device = dev or device()
        self.compute_dtype = compute_dtype
        self.vars = []
        self.params = self.grads = self.shadow = self.shadow_t = None
        self.size = 0

    def add(self, name, shape, init, decay=True, matrix=False):
        assert self.params is None, "arena already finalised"
        v = Variable(self, name, shape, decay, matrix, init)
        v.offset = self.size
        self.size += (v.size + ALIGN - 1) // ALIGN * ALIGN
        self.vars.append(v)
        return v

    def finalize(self):
        if self.params is not None:
            return self
        from . import ops
        # whole 256-byte lines per rank for every world size up to 8, and 16 (1680 = lcm(1..8, 16)): a gradient reduce-scatter
        # cuts the arena into `world` equal slices.  The unit does not depend on the world size the process happens
        # to run in (nor on whether comm was initialised before the model was built), so the arena of a model has ONE
        # size: a training state saved on N GPUs resumes on any other N.  `extent` = the end of the last variable.
        self.extent = self.size
        q = ALIGN * 1680
        self.size = (self.size + q - 1) // q * q
        host = np.zeros(self.size, np.float32)
        for v in self.vars:
            host[v.offset:v.offset + v.size] = np.asarray(v._init, np.float32).reshape(-1)
            v._init = None
        self.params = torch.from_numpy(host).to(self.device)
        self.grads = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        if self.compute_dtype == torch.bfloat16:
            self.shadow = torch.empty(self.size, dtype=torch.bfloat16, device=self.device)
            ops.cast(self.params, self.shadow)
            # transposed twin of every 2-D GEMM weight: dX = dY . W then reads W^T K-contiguously
            # (15-25 % faster than the K-strided operand path, tools/dx_bench.py); POLUS_DX_TRANSPOSED=0
            # keeps the single shadow
            mats = [v for v in self.vars if v.matrix and len(v.shape) == 2]
            if mats and os.environ.get("POLUS_DX_TRANSPOSED", "1") != "0":
                self.shadow_t = torch.zeros(self.size, dtype=torch.bfloat16, device=self.device)
                segs, t0 = [], 0
                for v in mats:
                    segs.append((v.offset, v.shape[0], v.shape[1], t0))
                    t0 += ((v.shape[0] + 63) // 64) * ((v.shape[1] + 63) // 64)
                self._tr_segs = torch.tensor(segs, dtype=torch.int64).to(self.device)
                self._tr_tiles = t0
                self.refresh_transposed()
        return self

    def refresh_transposed(self, var=None):
        """Re-derive the transposed bf16 shadows from the bf16 shadows (after every optimizer
        step / load / broadcast): 2 x 2 bytes per GEMM weight of extra HBM traffic."""
        if self.shadow_t is None:
            return
        from . import ops
        if var is None:
            ops.transpose_bf16_batched(self.shadow, self.shadow_t, self._tr_segs, self._tr_segs.shape[0], self._tr_tiles)
        elif var.matrix and len(var.shape) == 2:
            ops.transpose_bf16(var.compute, var.compute_t)

    def refresh_transposed_of(self, variables):
        """The transposed shadows of just these variables, in one launch: the in-backward optimizer update
        re-derives a window's right after updating it, instead of one launch for the whole arena after backward."""
        if self.shadow_t is None:
            return
        mats = [v for v in variables if v.matrix and len(v.shape) == 2]
        if not mats:
            return
        from . import ops
        cache = self.__dict__.setdefault("_tr_sub", {})
        key = tuple(id(v) for v in mats)
        t = cache.get(key)
        if t is None:
            segs, t0 = [], 0
            for v in mats:
                segs.append((v.offset, v.shape[0], v.shape[1], t0))
                t0 += ((v.shape[0] + 63) // 64) * ((v.shape[1] + 63) // 64)
            t = cache[key] = (torch.tensor(segs, dtype=torch.int64).to(self.device), t0)
        ops.transpose_bf16_batched(self.shadow, self.shadow_t, t[0], t[0].shape[0], t[1])

    def refresh_shadow(self, var=None):
        if self.shadow is None:
            return
        from . import ops
        if var is None:
            ops.cast(self.params, self.shadow)
            self.refresh_transposed()
            return
        else:
            n = (var.size + 3) // 4 * 4
            ops.cast(self.params[var.offset:var.offset + n], self.shadow[var.offset:var.offset + n])
        self.refresh_transposed(var)

    def zero_grads(self):
        self.grads.zero_()

    def state_dict(self):
        return {v.name: v.numpy() for v in self.vars}
