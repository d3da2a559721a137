"""One training step as a replayed HIP graph.

Small workloads (BASELINE.json configs[1]: BERT-base, 32 x 128 tokens) are launch-bound: a step is ~300 kernel
launches of a few microseconds each, every one issued from Python through ctypes.  After a few eager steps
`GraphedStep` captures `trainer._train_step` on a capture stream (torch.cuda.graph: hipStreamBeginCapture) and
from then on a step is: copy the batch into the static input tensors, write the step's three scalars into a
16-byte device block, replay.  What varies from step to step is read from memory, never from kernel arguments
(they are frozen in a replay): the dropout salt (polus_set_dynamic_params -- the kernels finish the seed mix
themselves, so eager and replayed steps draw identical masks), and AdamW's lr / lr_t.

Bit-for-bit the same training run as the eager path (tests/test_boundary_gpu.py::test_graphed_step_*).
Limits: one process (no data parallelism inside the capture), static batch shape, a model whose dropout
seeds come from `site_seed` (BertModel / TFBertSplited), a fused polus_amd optimizer, no gradient accumulation,
no post_process_grads that runs host code per step."""
import math
import struct

import torch

from . import _lib, comm, ops
from .models import dropout_salt
from .optimizers import Adam
from .tensor import DeviceScalar


def _leaves(x, prefix=""):
    if isinstance(x, dict):
        for k in sorted(x):
            yield from _leaves(x[k], f"{prefix}{k}.")
    elif isinstance(x, (tuple, list)):
        for i, v in enumerate(x):
            yield from _leaves(v, f"{prefix}{i}.")
    else:
        yield prefix, x


def _map_leaves(f, x):
    if isinstance(x, dict):
        return {k: _map_leaves(f, v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(_map_leaves(f, v) for v in x)
    return f(x)


def _reachable_tensors(roots, max_depth=8):
    """Every torch.Tensor reachable from `roots` through attributes, dicts, lists and tuples."""
    seen, out = set(), []
    stack = [(r, 0) for r in roots]
    while stack:
        obj, depth = stack.pop()
        if id(obj) in seen or obj is None:
            continue
        seen.add(id(obj))
        if torch.is_tensor(obj):
            out.append(obj)
            continue
        if depth >= max_depth or isinstance(obj, (str, bytes, int, float, bool, type)) or callable(obj) and not hasattr(obj, "__dict__"):
            continue
        if isinstance(obj, dict):
            stack.extend((v, depth + 1) for v in obj.values())
        elif isinstance(obj, (list, tuple, set)):
            stack.extend((v, depth + 1) for v in obj)
        elif hasattr(obj, "__dict__") and not isinstance(obj, type(torch)):
            stack.extend((v, depth + 1) for v in vars(obj).values())
    return out


class GraphedStep:
    RING = 32          # pinned staging slots for the dynamic block (a slot is reused only after its copy ran)

    def __init__(self, trainer, warmup=3):
        if comm.size() > 1:
            raise ValueError("GraphedStep: data-parallel steps are not captured (collectives stay outside the graph)")
        if not isinstance(trainer.optimizer, Adam):
            raise ValueError("GraphedStep needs a fused polus_amd optimizer (Adam / AdamWeightDecay)")
        if trainer.grad_accum_steps != 1 or trainer.post_process_grads is not None:
            raise ValueError("GraphedStep: gradient accumulation / post_process_grads are not supported")
        if not hasattr(trainer.model, "site_seed"):
            raise ValueError("GraphedStep: the model's dropout seeds must come from site_seed (BertModel)")
        self.trainer, self.warmup = trainer, int(warmup)
        self.seen, self.graph, self.key = 0, None, None
        dev = trainer.model.arena.device
        self.block = torch.zeros(4, dtype=torch.int32, device=dev)
        self.stage = torch.zeros((self.RING, 4), dtype=torch.int32).pin_memory()
        self.stage_np = self.stage.numpy()
        self.events = [None] * self.RING
        self.slot = 0
        self.static_in = None
        self.loss_ring = torch.empty(256, dtype=torch.float32, device=dev)
        self.loss_k = 0

    # ---- per-step scalars
    def _write_block(self, salt, lr, lr_t):
        k = self.slot
        self.slot = (k + 1) % self.RING
        if self.events[k] is not None:
            self.events[k].synchronize()               # the copy that last read this slot is done
        self.stage_np[k] = struct.unpack("4i", struct.pack("Iffi", salt & 0xFFFFFFFF, lr, lr_t, 0))
        self.block.copy_(self.stage[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev

    def _signature(self, inputs):
        return tuple((name, tuple(t.shape), str(t.dtype)) for name, t in _leaves(inputs) if torch.is_tensor(t))

    def _to_static(self, inputs):
        dev = self.trainer.model.arena.device

        def conv(a):
            t = a if torch.is_tensor(a) else torch.as_tensor(a)
            if t.dtype == torch.int64:
                t = t.to(torch.int32)
            return t.to(dev)
        return _map_leaves(conv, inputs)

    def _capture(self, inputs):
        tr, model, opt = self.trainer, self.trainer.model, self.trainer.optimizer
        self.static_in = _map_leaves(lambda t: t.clone(), inputs)
        self.key = self._signature(inputs)
        overlap = getattr(model, "overlap_dw", False)
        model.overlap_dw = False                       # one stream inside the capture
        model.graph_seeds = True
        ops.set_dynamic_params(self.block)
        # one scratch buffer for the whole captured step, as large as any stream's scratch grew during the eager
        # steps (the weight-gradient launches ran on a side stream there): no growth -- a sync -- inside the capture
        ws = ops.workspace(self.trainer.model.arena.device)
        ws.get(max([w.buf.numel() for w in ops._WS.values() if w.buf is not None] + [1 << 20]))
        ops.WORKSPACE_OVERRIDE = ws
        # the state a step advances on the host: the capture must not move it (the replay bookkeeping does)
        it, ds, mc = opt.iterations, model.dropout_step, getattr(tr, "step_counter_micro", 0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                with _lib.pinned_stream():
                    loss = tr._train_step(*self.static_in)
        finally:
            ops.WORKSPACE_OVERRIDE = None
            ops.set_dynamic_params(None)
            model.graph_seeds = False
            model.overlap_dw = overlap
            opt.iterations, model.dropout_step, tr.step_counter_micro = it, ds, mc
        self.graph, self.captured_loss = g, loss.t
        # The graph holds RAW POINTERS to every buffer the step touched: the shape-keyed activation / scratch caches of
        # the model and its layers, the loss buffers, the optimizer slots, the workspace.  A later eager step of another
        # batch shape (a short last batch) or an inference call between epochs replaces those cache entries and may grow
        # the workspace; without these references the old blocks would return to the caching allocator and the next replay
        # would read and write memory that other tensors own by then.
        self._keepalive = _reachable_tensors([tr.model, tr.loss, opt, ws, self.static_in, list(ops._WS.values())])
        torch.cuda.synchronize()

    def __call__(self, *inputs):
        tr, model, opt = self.trainer, self.trainer.model, self.trainer.optimizer
        if ops.GEMM_PROFILE is not None or self.seen < self.warmup:
            self.seen += 1
            return tr._eager_step(*inputs)
        inputs = self._to_static(inputs)
        if self.graph is None:
            self._capture(inputs)
        elif self._signature(inputs) != self.key:      # another batch shape: that step runs eagerly
            return tr._eager_step(*inputs)
        for (_, dst), (_, src) in zip(_leaves(self.static_in), _leaves(inputs)):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        # host bookkeeping of the step the graph is about to run (what _train_step would have done)
        lr = opt.learning_rate(opt.iterations)
        opt.iterations += 1
        t = opt.iterations
        lr_t = lr * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)
        self._write_block(dropout_salt(model.dropout_step), lr, lr_t)
        model.dropout_step += 1
        tr.step_counter_micro = getattr(tr, "step_counter_micro", 0) + 1
        ops.set_dynamic_params(self.block)
        try:
            self.graph.replay()
        finally:
            ops.set_dynamic_params(None)
        k = self.loss_k
        self.loss_k = (k + 1) % self.loss_ring.numel()
        out = self.loss_ring[k:k + 1]
        out.copy_(self.captured_loss)
        return DeviceScalar(out)
