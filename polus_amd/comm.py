"""Data-parallel communication for the Polus step: RCCL over xGMI, one process per GPU.

Replaces the six Horovod touch points of the reference (polus/mock/horovod.py:5-24 lists the
surface; call sites polus/__init__.py:109-122, polus/training.py:182,210-211,
polus/callbacks.py:249).  Two planes:

* control plane (host objects, barriers, the 128-byte RCCL id): a `gloo` process group over the
  torchrun rendezvous -- plumbing;
* data plane (gradients, parameters): RCCL.  By default through torch.distributed's binding (`nccl`);
  `POLUS_DIST_BACKEND=native` selects the `polus_comm_*` entry points of libpolus_hip.so (include/polus_hip.h: the
  same RCCL collectives through the C ABI, queued on a side HIP stream and fenced against the compute stream with
  events) -- opt-in until a multi-GPU run has pinned it; if it cannot be brought up on some rank, every rank falls
  back to the torch binding together (the ranks vote over the control plane before any of them enters
  ncclCommInitRank).  `gloo` is the CPU transport (the multi-process tests, and several ranks sharing one GPU --
  RCCL refuses duplicate devices).

Horovod's DistributedGradientTape averages every gradient tensor with one all-reduce per tensor fused
by a background thread.  Here the gradients already live in one flat f32 arena laid out in forward
order, so `GradBucketReducer` cuts it into contiguous buckets from the END (the order backward produces
them) and starts a bucket's collective as soon as the model reports everything above its lower edge
final.  Two exchange schemes:

* `allreduce` (default): all-reduce (SUM) per bucket, buckets cut at variable boundaries; `on_launched` lets the
  trainer queue the AdamW update of a bucket's variables behind that bucket's completion event, so exchange and
  update both run beside the rest of backward (polus_amd/training.py _UpdateBehindAllReduce);
* `rs` (`POLUS_DP_MODE=rs`): reduce-scatter per bucket -- rank r ends up with the sum of slice r of every
  bucket -- each rank runs AdamW on its slices only (1/N of the optimizer's HBM traffic), then one all-gather
  per bucket returns the updated f32 parameters to everyone.  Half the bytes during backward, but the all-gather
  can only follow the update and nothing hides it (DESIGN.md §5 has the arithmetic behind the default).

Optional bf16 transport of the gradients (`POLUS_DP_BF16=1`, either scheme) halves the bytes on the wire; the sums are
then rounded to bf16.

The 1/world (and 1/accumulation) factor is folded into the optimizer's gradient scale instead of a
separate pass."""
import ctypes
import logging
import os

import torch
import torch.distributed as dist

_STATE = {"initialized": False, "world": 1, "rank": 0, "local_rank": 0, "backend": None, "plane": None}


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


# ------------------------------------------------------------------------------------ data planes
class _Done:
    """A collective that is already ordered with respect to its consumer."""

    def wait(self):
        pass


class _TorchPlane:
    """torch.distributed collectives on a process group (gloo on CPU, nccl = RCCL on GPUs)."""
    name = "torch"

    def __init__(self, group=None):
        self.group = group

    def all_reduce_sum(self, t):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_scatter_sum(self, send, recv):
        return dist.reduce_scatter_tensor(recv, send, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_gather(self, send, recv):
        return dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)

    def broadcast(self, t, root):
        dist.broadcast(t, src=root, group=self.group)

    def close(self):
        pass


class _GlooPlane(_TorchPlane):
    """gloo has no reduce_scatter_tensor / all_gather_into_tensor on every build: spell them with all_reduce /
    all_gather (tests only; the byte counts of the real plane do not matter here)."""
    name = "gloo"

    def reduce_scatter_sum(self, send, recv):
        tmp = send.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
        n = recv.numel()
        recv.copy_(tmp[rank() * n:(rank() + 1) * n])
        return _Done()

    def all_gather(self, send, recv):
        n = send.numel()
        parts = [torch.empty_like(send) for _ in range(size())]
        dist.all_gather(parts, send.contiguous(), group=self.group)
        for r, p in enumerate(parts):
            recv[r * n:(r + 1) * n].copy_(p)
        return _Done()


class _StreamHandle:
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class _NativePlane:
    """polus_comm_* (RCCL through the C ABI) on a side HIP stream.  Every call first makes that stream wait
    for what the compute stream has queued so far; the handle's wait() makes the compute stream wait for
    the collective."""
    name = "native"

    def __init__(self, world, rk):
        # Every rank reaches every control-plane collective below whatever failed locally, and the ranks vote BEFORE
        # any of them enters ncclCommInitRank: a rank that raised early (library missing, symbol missing, no unique
        # id) would otherwise leave its peers blocked inside the RCCL bootstrap, which has no timeout.
        err, uid = None, (ctypes.c_ubyte * 128)()
        try:
            from . import _lib
            # the RCCL copy that shares torch's HIP runtime (same SONAME as /opt/rocm's)
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(bundled):
                ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
            self.lib = _lib.load()
            self.check = _lib.check
            self.check(self.lib.polus_comm_group_start(), "polus_comm_group_start")     # dlopen + every RCCL symbol
            self.check(self.lib.polus_comm_group_end(), "polus_comm_group_end")
            if rk == 0:
                self.check(self.lib.polus_comm_unique_id(uid), "polus_comm_unique_id")
        except Exception as e:      # noqa: BLE001
            err = e
        box = [bytes(uid) if (rk == 0 and err is None) else None]
        dist.broadcast_object_list(box, src=0)            # control plane (gloo)
        ready = [None] * world
        dist.all_gather_object(ready, err is None and box[0] is not None)
        if not all(ready):
            raise RuntimeError(f"native RCCL plane: local bring-up failed on rank(s) {[r for r, ok in enumerate(ready) if not ok]}"
                               + (f" ({err})" if err is not None else ""))
        uid = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        self.comm = ctypes.c_void_p()
        self.check(self.lib.polus_comm_init(ctypes.byref(self.comm), rk, world, uid), "polus_comm_init")
        self.stream = torch.cuda.Stream()
        self.world, self.rank = world, rk

    def info(self):
        """(ranks, rank, device) as RCCL reports them for this communicator."""
        n, r, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.check(self.lib.polus_comm_info(self.comm, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "polus_comm_info")
        return n.value, r.value, d.value

    def _enter(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)

    def _leave(self):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _StreamHandle(ev)

    @staticmethod
    def _dt(t):
        from ._lib import dtype_code
        return dtype_code(t.dtype)

    def all_reduce_sum(self, t):
        self._enter()
        self.check(self.lib.polus_comm_allreduce_sum(self.comm, t.data_ptr(), t.numel(), self._dt(t), self.stream.cuda_stream),
                   "polus_comm_allreduce_sum")
        return self._leave()

    def reduce_scatter_sum(self, send, recv):
        assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype
        self._enter()
        self.check(self.lib.polus_comm_reduce_scatter_sum(self.comm, send.data_ptr(), recv.data_ptr(), recv.numel(),
                                                          self._dt(send), self.stream.cuda_stream), "polus_comm_reduce_scatter_sum")
        return self._leave()

    def all_gather(self, send, recv):
        assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
        self._enter()
        self.check(self.lib.polus_comm_all_gather(self.comm, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                  self._dt(send), self.stream.cuda_stream), "polus_comm_all_gather")
        return self._leave()

    def broadcast(self, t, root):
        self._enter()
        self.check(self.lib.polus_comm_broadcast(self.comm, t.data_ptr(), t.numel() * t.element_size(), int(root),
                                                 self.stream.cuda_stream), "polus_comm_broadcast")
        self._leave().wait()

    def close(self):
        if self.comm:
            torch.cuda.synchronize()
            self.lib.polus_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()


def _bring_up_native(world, rk):
    """Native communicator + a one-element all-reduce as a self-test; every rank learns whether ALL succeeded."""
    plane, err = None, None
    try:
        plane = _NativePlane(world, rk)
        probe = torch.ones(4, dtype=torch.float32, device="cuda")
        plane.all_reduce_sum(probe).wait()
        torch.cuda.synchronize()
        if not torch.equal(probe.cpu(), torch.full((4,), float(world))):
            raise RuntimeError(f"self-test all-reduce returned {probe.tolist()} for world size {world}")
    except Exception as e:      # noqa: BLE001 -- any failure means: use the other binding, together
        err = e
    oks = [None] * world
    dist.all_gather_object(oks, err is None)
    if all(oks):
        return plane
    if plane is not None:
        try:
            plane.close()
        except Exception:       # noqa: BLE001
            pass
    from .context import logger
    logger.warning(f"native RCCL plane unavailable on rank(s) {[r for r, ok in enumerate(oks) if not ok]} "
                   f"({err}); falling back to torch.distributed's RCCL binding")
    return None


def init():
    """hvd.init(): joins the torchrun rendezvous when WORLD_SIZE > 1; otherwise the
    world_size-1 behaviour of polus/mock/horovod.py ("mock")."""
    if _STATE["initialized"]:
        return _STATE["backend"] or "mock"
    world = _env_int("WORLD_SIZE", 1)
    _STATE["initialized"] = True
    if world <= 1:
        return "mock"
    rk, local_rank = _env_int("RANK", 0), _env_int("LOCAL_RANK", 0)
    use_gpu = torch.cuda.is_available()
    # Data plane on GPUs: torch.distributed's RCCL binding by default.  The native plane (polus_comm_* through the C ABI,
    # POLUS_DIST_BACKEND=native) issues the same RCCL collectives on a stream of ours; it has been driven end to end on
    # a communicator of size 1 only (no multi-GPU box in the build environment), so it stays opt-in until a run on
    # >= 2 GPUs has recorded parity and timing for it -- "multi-GPU parity unpinned", DESIGN.md section 5.
    backend = os.environ.get("POLUS_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if use_gpu:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if use_gpu and backend != "gloo":
        # RCCL's channel kernels are persistent workgroups, one CU each, for as long as a collective runs.  The GEMMs of the
        # step keep ONE 512-thread, 128 KiB workgroup per CU (all of a CU's registers and LDS), so a CU that runs a channel
        # takes no tile: cap the channels (POLUS_RCCL_MAX_CHANNELS, default 32; an explicit NCCL_MAX_NCHANNELS wins) and let
        # the GEMM tile-shape choice plan for that many CUs fewer (POLUS_GEMM_RESERVE_CUS; csrc/gemm.hip pp_tile), so that a
        # 256-tile launch does not find 240 CUs and run two rounds.  The exchange has ~8 ms of backward to hide under and
        # needs ~60 GB/s per direction and link for that.  32 rather than 16: every launch of the step has the same number of
        # tile rounds on 224 CUs as on 240 (192-, 216-, 576- and 768-tile launches), so the headroom for the exchange is free.  The reserve is switched on only around
        # backward (training.py _train_step): the forward pass runs beside no collective and keeps all 256 CUs.
        cap = os.environ.get("POLUS_RCCL_MAX_CHANNELS", "32")
        if "NCCL_MAX_NCHANNELS" not in os.environ:
            os.environ["NCCL_MAX_NCHANNELS"] = cap
            # (process-wide: every other RCCL user of this process is capped as well -- say so once)
            logging.getLogger("polus_amd").info("comm.init: NCCL_MAX_NCHANNELS=%s set for this process (POLUS_RCCL_MAX_CHANNELS; an explicit "
                                                "NCCL_MAX_NCHANNELS in the environment is kept)", cap)
        if "POLUS_GEMM_RESERVE_CUS" not in os.environ:
            os.environ["POLUS_GEMM_RESERVE_CUS"] = os.environ["NCCL_MAX_NCHANNELS"]
            try:
                from . import _lib
                _lib.check(_lib.load().polus_reload_env(), "polus_reload_env")
                _lib.check(_lib.load().polus_set_reserve_active(0), "polus_set_reserve_active")
            except Exception:       # noqa: BLE001 -- the library is loaded (and reads its switches) later in that case
                pass
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rk, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend="gloo", rank=rk, world_size=world)
    _STATE.update(world=dist.get_world_size(), rank=dist.get_rank(), local_rank=local_rank)
    plane = None
    if backend == "native":
        plane = _bring_up_native(world, rk)
        if plane is None:
            backend = "nccl"
            plane = _TorchPlane(dist.new_group(backend="nccl"))
    elif backend == "nccl":
        plane = _TorchPlane(None)
    else:
        plane = _GlooPlane(None)
    _STATE.update(backend=backend, plane=plane)
    return backend


def shutdown():
    plane = _STATE.get("plane")
    if plane is not None:
        plane.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    _STATE.update(initialized=False, world=1, rank=0, local_rank=0, backend=None, plane=None)


def size():
    return _STATE["world"]


def rank():
    return _STATE["rank"]


def local_rank():
    """The reference shards data and gates rank-0 side effects on local_rank()
    (polus/data.py:96, polus/callbacks.py:27) — single-node semantics, kept."""
    return _STATE["local_rank"]


def is_distributed():
    return _STATE["world"] > 1


def data_plane():
    return _STATE["plane"]


def data_plane_info():
    """What the data plane itself reports (a collective: every rank calls it): the sum of a one-per-rank all-reduce
    pushed through the plane -- the number of ranks the gradient exchange really spans -- and, on the native plane,
    RCCL's own ncclCommCount / ncclCommUserRank / ncclCommCuDevice.  bench.py prints it as `rccl_ranks`."""
    out = {"data_plane": _STATE["backend"] or "none", "world": size(), "ranks_seen": 1}
    plane = data_plane()
    if plane is None:
        return out
    dev = "cuda" if (torch.cuda.is_available() and _STATE["backend"] != "gloo") else None
    if dev is None and torch.cuda.is_available():
        dev = "cuda"
    probe = torch.ones(64, dtype=torch.float32, device=dev or "cpu")
    w = plane.all_reduce_sum(probe)
    if w is not None:
        w.wait()
    if probe.is_cuda:
        torch.cuda.synchronize()
    out["ranks_seen"] = int(round(float(probe[0].item())))
    if isinstance(plane, _NativePlane):
        n, r, d = plane.info()
        out.update(rccl_comm_count=n, rccl_comm_rank=r, rccl_comm_device=d)
    return out


def DistributedGradientTape(tape):
    """API-compat shim: there is no tape; gradient averaging is done by GradBucketReducer."""
    return tape


def _flat_tensors(variables):
    """Variables of one arena -> [arena.params]; raw tensors pass through."""
    out, seen = [], set()
    for v in variables:
        if hasattr(v, "arena"):
            if id(v.arena) not in seen:
                seen.add(id(v.arena))
                out.append(("arena", v.arena))
        else:
            out.append(("tensor", v))
    return out


def broadcast_variables(variables, root_rank=0):
    """hvd.broadcast_variables: one flat broadcast per arena (437.9 MB for BERT-base)
    instead of one per variable; refreshes the bf16 shadow afterwards."""
    if not is_distributed():
        return
    plane = data_plane()
    for kind, obj in _flat_tensors(variables):
        if kind == "arena":
            plane.broadcast(obj.params, root_rank)
            obj.refresh_shadow()
        else:
            plane.broadcast(obj, root_rank)


def allgather_object(y):
    """hvd.allgather_object (polus/callbacks.py:249): list with one entry per rank (control plane)."""
    if not is_distributed():
        return [y]
    out = [None] * size()
    if torch.is_tensor(y):
        y = y.detach().cpu()
    elif isinstance(y, (tuple, list)):
        y = type(y)(t.detach().cpu() if torch.is_tensor(t) else t for t in y)
    dist.all_gather_object(out, y)
    return out


def barrier():
    if is_distributed():
        dist.barrier()


def max_over_ranks(value):
    """Host scalar -> its maximum over the ranks (bench.py's timing rule)."""
    if not is_distributed():
        return value
    vals = [None] * size()
    dist.all_gather_object(vals, float(value))
    return max(vals)


class GradBucketReducer:
    """Bucketed, backward-overlapped gradient exchange over a flat arena (see the module docstring).

    mode "allreduce": buckets cut at the given tensor `boundaries` (sorted offsets), ~bucket_bytes each.
    mode "rs": buckets of equal size (a multiple of 64 * world elements, so that every slice is whole
    256-byte lines), the front bucket takes the remainder; `grads.numel()` must be a multiple of 64 * world
    (ParamArena pads to 64 * 1680 elements: every world size up to 8, and 16; other sizes fall back to all-reduce with a warning).  `transport_dtype=torch.bfloat16` sends the gradients as bf16."""

    def __init__(self, grads, bucket_bytes=64 << 20, boundaries=None, mode="allreduce", transport_dtype=None, plane=None):
        self.grads = grads
        self.mode = mode
        self.plane = plane or data_plane()
        self.world, self.rank = size(), rank()
        n = grads.numel()
        elems = max(1, bucket_bytes // grads.element_size())
        if mode == "rs":
            q = 64 * self.world
            assert n % q == 0, f"arena of {n} elements is not a multiple of 64 x world = {q}"
            step = max(q, elems // q * q)
            buckets, hi = [], n
            while hi > 0:
                lo = hi - step if hi - step >= step // 2 else 0      # no tiny front bucket
                lo = max(lo, 0)
                buckets.append((lo, hi))
                hi = lo
        else:
            # cut at tensor boundaries when given (sorted offsets), from the end of the arena
            cuts = sorted(set(boundaries or [])) or list(range(0, n, elems))
            cuts = [c for c in cuts if 0 <= c < n]
            if not cuts or cuts[0] != 0:
                cuts = [0] + cuts
            # a tensor larger than a bucket (the word-embedding table: 89 MB of gradient against 64 MB buckets) is cut
            # into equal parts of whole 256-byte lines: its first part's all-reduce is on the wire while the update of
            # nothing but the LAST part stays exposed behind it, instead of one 89 MB collective followed by one
            # 89 MB optimizer sweep at the very end of the step
            fine = []
            for lo_c, hi_c in zip(cuts, cuts[1:] + [n]):
                fine.append(lo_c)
                if hi_c - lo_c > elems:
                    parts = -(-(hi_c - lo_c) // elems)
                    step = -(-(hi_c - lo_c) // parts)
                    step = (step + 511) // 512 * 512
                    fine += [c for c in range(lo_c + step, hi_c, step)]
            cuts = fine
            buckets, hi = [], n
            lo_idx = len(cuts) - 1
            while hi > 0:
                lo = cuts[lo_idx]
                while lo_idx > 0 and hi - cuts[lo_idx - 1] <= elems:
                    lo_idx -= 1
                    lo = cuts[lo_idx]
                buckets.append((lo, hi))
                hi = lo
                lo_idx -= 1
        self.buckets = buckets          # descending offsets
        self.transport_dtype = transport_dtype if (transport_dtype is not None and transport_dtype != grads.dtype) else None
        self._stage = None
        self._next = 0
        self._ready_lo = n
        self._works = []
        self.launched_bytes = 0
        self.on_launched = None         # optional callback(lo, hi, work) right after a bucket's collective is queued

    # ---- slices this rank owns after the reduce-scatter: slice `rank` of every bucket
    def owned_ranges(self):
        out = []
        for lo, hi in self.buckets:
            s = (hi - lo) // self.world
            out.append((lo + self.rank * s, lo + (self.rank + 1) * s))
        return sorted(out)

    def begin(self):
        self._next, self._ready_lo, self._works = 0, self.grads.numel(), []

    def on_ready(self, lo, hi, variables=None):
        """Model hook: grads[lo:hi] are final (called in descending arena order)."""
        self._ready_lo = min(self._ready_lo, lo)
        self._launch_ready()

    def _launch_one(self, lo, hi):
        view = self.grads[lo:hi]
        if self.mode != "rs":
            if self.transport_dtype is None:
                self.launched_bytes += view.numel() * view.element_size()
                return self.plane.all_reduce_sum(view)
            # bf16 on the wire: cast the bucket, all-reduce the copy (sums rounded to bf16), widen it back
            from . import ops
            if self._stage is None:
                self._stage = torch.empty(self.grads.numel(), dtype=self.transport_dtype, device=self.grads.device)
            st = self._stage[lo:hi]
            ops.cast(view, st)
            self.launched_bytes += st.numel() * st.element_size()
            return _CastBack(self.plane.all_reduce_sum(st), st, view)
        s = (hi - lo) // self.world
        mine = self.grads[lo + self.rank * s:lo + (self.rank + 1) * s]
        if self.transport_dtype is None:
            self.launched_bytes += view.numel() * view.element_size()
            return self.plane.reduce_scatter_sum(view, mine)       # in place: recv is slice `rank` of send
        # bf16 on the wire: cast the bucket, reduce-scatter the copy, cast the owned slice back
        from . import ops
        if self._stage is None:
            self._stage = torch.empty(self.grads.numel(), dtype=self.transport_dtype, device=self.grads.device)
        st = self._stage[lo:hi]
        ops.cast(view, st)
        st_mine = self._stage[lo + self.rank * s:lo + (self.rank + 1) * s]
        self.launched_bytes += st.numel() * st.element_size()
        h = self.plane.reduce_scatter_sum(st, st_mine)
        return _CastBack(h, st_mine, mine)

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= self._ready_lo:
            lo, hi = self.buckets[self._next]
            # async: the comm stream waits for the kernels queued so far on the compute stream, then
            # reduces beside whatever backward launches next
            self._works.append(self._launch_one(lo, hi))
            self._next += 1
            if self.on_launched is not None:
                self.on_launched(lo, hi, self._works[-1])

    def finish(self, keep_last=False):
        """Flush what is left and make the current stream wait for every bucket.
        keep_last (allreduce mode): leave the last bucket (the front of the arena: the embeddings, whose
        gradient is the last thing backward produces) in flight and return its upper offset -- the caller
        updates the parameters above it first and then calls finish_last().  Returns None when there is
        nothing to keep (a single bucket)."""
        self._ready_lo = 0
        self._launch_ready()
        keep = keep_last and self.mode != "rs" and len(self._works) > 1 and self._next == len(self.buckets)
        for w in (self._works[:-1] if keep else self._works):
            w.wait()
        self._works = self._works[-1:] if keep else []
        return self.buckets[-1][1] if keep else None

    def finish_last(self):
        for w in self._works:
            w.wait()
        self._works = []

    def allgather(self, params):
        """rs mode, after the sharded optimizer step: every rank's updated slices back into the full arena."""
        assert self.mode == "rs" and params.numel() == self.grads.numel()
        works = []
        for lo, hi in self.buckets:
            s = (hi - lo) // self.world
            works.append(self.plane.all_gather(params[lo + self.rank * s:lo + (self.rank + 1) * s], params[lo:hi]))
        for w in works:
            w.wait()


class _CastBack:
    """wait() of a bf16-transported bucket: after the collective, widen the reduced (owned) part back into the f32 arena."""

    def __init__(self, handle, src, dst):
        self.handle, self.src, self.dst = handle, src, dst
        self._widened = False

    def wait(self):
        """The first wait() widens on the stream it is called from; later ones (the reducer's finish() after the
        trainer has already queued the bucket's update behind it) only wait for the collective."""
        from . import ops
        self.handle.wait()
        if not self._widened:
            ops.cast(self.src, self.dst)
            self._widened = True
