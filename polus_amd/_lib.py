"""ctypes binding of libpolus_hip.so (include/polus_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this
module raises.  PyTorch is used by callers only to own device memory and streams; every
pointer handed over here is a raw device address.
"""
import ctypes
import os

from . import build as _build

F32, BF16 = 0, 1
K_CONTIG, K_STRIDED = 0, 1
ACT_NONE, ACT_GELU, ACT_SWISH, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4
GEMM_ACCUM_C, GEMM_ACT_FWD, GEMM_ACT_BWD = 1, 2, 4

_c = ctypes
_vp, _i, _l, _f, _sz, _i64, _u32 = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float, _c.c_size_t, _c.c_int64, _c.c_uint32

# name -> (restype, argtypes); mirrors include/polus_hip.h one to one
SIGNATURES = {
    "polus_last_error": (_c.c_char_p, []),
    "polus_abi_version": (_i, []),
    "polus_device_info": (_i, [_c.POINTER(_i), _c.POINTER(_i), _c.c_char_p, _i]),
    "polus_reload_env": (_i, []),
    "polus_set_reserve_active": (_i, [_i]),
    "polus_set_dynamic_params": (_i, [_vp]),
    "polus_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "polus_gemm_auto_split": (_i, [_i, _i, _i]),
    "polus_gemm": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _f,
                        _vp, _vp, _l, _vp, _l, _i, _i, _i, _vp, _sz, _vp]),
    "polus_gemm_dropout": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _f,
                                _vp, _vp, _l, _vp, _l, _i, _i, _i, _vp, _sz, _f, _u32, _vp]),
    "polus_dropout_mask": (_i, [_u32, _f, _u32, _i64, _vp, _vp]),
    "polus_dropout": (_i, [_i, _vp, _vp, _i64, _f, _u32, _vp]),
    "polus_dense_bwd_params_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "polus_dense_bwd_params": (_i, [_i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "polus_attention_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u32, _vp]),
    "polus_attention_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "polus_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u32, _vp, _sz, _vp]),
    "polus_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "polus_layernorm_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "polus_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _u32, _vp, _sz, _vp]),
    "polus_layernorm_bwd_finalize": (_i, [_vp, _sz, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "polus_embed_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "polus_embed_ln_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i, _i, _i, _i, _i, _i, _f, _f, _u32, _vp]),
    "polus_embed_ln_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _i, _i,
                                _i, _i, _i, _i, _i, _i, _f, _u32, _vp, _sz, _vp]),
    "polus_colsum_workspace_bytes": (_sz, [_i, _i]),
    "polus_colsum": (_i, [_i, _vp, _l, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "polus_loss_workspace_bytes": (_sz, [_i]),
    "polus_softmax_xent": (_i, [_i, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _sz, _vp]),
    "polus_sigmoid_xent": (_i, [_i, _vp, _l, _vp, _l, _vp, _f, _vp, _vp, _l, _i, _i, _vp, _sz, _vp]),
    "polus_crf_workspace_bytes": (_sz, [_i, _i, _i]),
    "polus_crf_nll": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "polus_crf_viterbi": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "polus_confusion_matrix": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "polus_argmax": (_i, [_vp, _l, _vp, _i, _i, _vp]),
    "polus_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64,
                             _f, _f, _f, _f, _f, _f, _f, _vp, _vp]),
    "polus_sqnorm_workspace_bytes": (_sz, [_i64]),
    "polus_sqnorm": (_i, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "polus_sqnorm_segments": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "polus_clip_scale": (_i, [_vp, _i, _f, _f, _vp, _vp]),
    "polus_cast": (_i, [_i, _vp, _i, _vp, _i64, _vp]),
    "polus_scale": (_i, [_vp, _f, _i64, _vp]),
    "polus_act_bwd": (_i, [_i, _vp, _vp, _vp, _i64, _i, _vp]),
    "polus_transpose_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "polus_transpose_bf16_batched": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "polus_dense_bwd_params_grouped_workspace_bytes": (_sz, [_i, _vp, _i, _i]),
    "polus_dense_bwd_params_grouped": (_i, [_i, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "polus_dense_thin_supported": (_i, [_i, _i, _i]),
    "polus_dense_thin_fwd": (_i, [_i, _vp, _l, _vp, _l, _vp, _i, _vp, _l, _i, _i, _i, _vp]),
    "polus_dense_thin_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "polus_dense_thin_bwd": (_i, [_i, _vp, _l, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "polus_comm_unique_id": (_i, [_vp]),
    "polus_comm_init": (_i, [_c.POINTER(_vp), _i, _i, _vp]),
    "polus_comm_destroy": (_i, [_vp]),
    "polus_comm_info": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "polus_comm_broadcast": (_i, [_vp, _vp, _sz, _i, _vp]),
    "polus_comm_allreduce_sum": (_i, [_vp, _vp, _sz, _i, _vp]),
    "polus_comm_reduce_scatter_sum": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "polus_comm_all_gather": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "polus_comm_group_start": (_i, []),
    "polus_comm_group_end": (_i, []),
}


class DwProblem(ctypes.Structure):
    """struct polus_dw_problem of include/polus_hip.h"""
    _fields_ = [("dY", ctypes.c_void_p), ("lddy", ctypes.c_long), ("X", ctypes.c_void_p), ("ldx", ctypes.c_long),
                ("dW", ctypes.c_void_p), ("lddw", ctypes.c_long), ("db", ctypes.c_void_p),
                ("n_out", ctypes.c_int), ("n_in", ctypes.c_int)]


class PolusHipError(RuntimeError):
    pass


_LIB = None


def load():
    """Loads (never builds) the in-tree shared library; raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.lib_path()
    if not os.path.exists(path):
        raise PolusHipError(
            f"{path} is missing: build it with `python -m polus_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback for the training path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  Load torch's copy first so this library binds to the runtime that owns
    # torch's allocations and streams; loading ours first would start a second runtime.
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.polus_abi_version() != 1:
        raise PolusHipError("libpolus_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().polus_last_error()
        raise PolusHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# torch.cuda.current_stream() costs ~8 us per call and a training step makes ~270 of them (one
# per kernel launch): inside a `pinned_stream()` scope the handle is looked up once.  The scope is
# entered by the trainer around a step; code that switches streams inside it does so through
# `stream_scope`, which keeps the cached handle in step with torch's current stream.
_CUR_STREAM = [None]


def current_stream():
    h = _CUR_STREAM[0]
    if h is not None:
        return h
    import torch
    return torch.cuda.current_stream().cuda_stream


class pinned_stream:
    """with pinned_stream(): kernel launches inside use the stream that is current at entry."""

    def __enter__(self):
        import torch
        self._prev = _CUR_STREAM[0]
        if self._prev is None and torch.cuda.is_available():
            _CUR_STREAM[0] = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _CUR_STREAM[0] = self._prev
        return False


class stream_scope:
    """with stream_scope(s): torch.cuda.stream(s) plus the cached handle."""

    def __init__(self, stream):
        import torch
        self._stream = stream
        self._ctx = torch.cuda.stream(stream)

    def __enter__(self):
        self._prev = _CUR_STREAM[0]
        self._ctx.__enter__()
        if self._prev is not None:
            _CUR_STREAM[0] = self._stream.cuda_stream
        return self

    def __exit__(self, *exc):
        _CUR_STREAM[0] = self._prev
        return self._ctx.__exit__(*exc)


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise PolusHipError(f"unsupported dtype {torch_dtype}")
