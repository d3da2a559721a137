"""Host-side launchers: torch tensors in, raw pointers through the C ABI, nothing computed here.

torch is the allocator and the stream owner only.  Every function launches on torch's
current stream and returns without synchronising.
"""
import torch

from . import _lib
from ._lib import (F32, BF16, K_CONTIG, K_STRIDED, ACT_NONE, ACT_GELU, ACT_SWISH, ACT_RELU, ACT_TANH,
                   GEMM_ACCUM_C, GEMM_ACT_FWD, GEMM_ACT_BWD, check, ptr, dtype_code)

# bench.py instrumentation: when a list, every gemm launch appends (kind, flops, ev0, ev1)
GEMM_PROFILE = None

ACT_CODES = {None: ACT_NONE, "linear": ACT_NONE, "gelu": ACT_GELU, "swish": ACT_SWISH,
             "relu": ACT_RELU, "tanh": ACT_TANH}


class Workspace:
    """Grow-only device scratch. Kernels on one stream run in order, so one buffer per
    stream is enough; it is sized during the first step and never reallocated after."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            # drop the old buffer only after queued kernels are done with it
            if self.buf is not None:
                torch.cuda.current_stream().synchronize()
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self.buf


_WS = {}
WORKSPACE_OVERRIDE = None      # polus_amd/graph.py: the scratch a captured step uses (same buffer in every replay)


_HAS_GPU = None


def workspace(device):
    global _HAS_GPU
    if WORKSPACE_OVERRIDE is not None:
        return WORKSPACE_OVERRIDE
    if _HAS_GPU is None:
        _HAS_GPU = torch.cuda.is_available()      # ~20 us per call otherwise, once per launch
    key = (device, _lib.current_stream() if _HAS_GPU else 0)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = Workspace(device)
    return ws


def _st():
    return _lib.current_stream()


def set_dynamic_params(block):
    """Register (None: unregister) the 16-byte device block {uint32 salt, f32 lr, f32 lr_t, 0} that dropout
    kernels and the fused optimizer read per-step scalars from (include/polus_hip.h)."""
    if block is not None:
        _req_cuda(block)
        assert block.numel() * block.element_size() >= 16 and block.is_contiguous()
    check(_lib.load().polus_set_dynamic_params(ptr(block)), "polus_set_dynamic_params")


def set_env(name, value=None):
    """Set (or, with None, unset) one of the library's POLUS_* tuning switches and make the library
    re-read them: they are cached at load time (include/polus_hip.h polus_reload_env)."""
    import os
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    check(_lib.load().polus_reload_env(), "polus_reload_env")
    _AUTO_SPLIT.clear()


def reserve_cus(active):
    """Switch the CU reserve of the GEMM launches (POLUS_GEMM_RESERVE_CUS) on or off: on while RCCL's channel kernels
    share the chip (backward of a data-parallel step), off otherwise (include/polus_hip.h polus_set_reserve_active)."""
    check(_lib.load().polus_set_reserve_active(1 if active else 0), "polus_set_reserve_active")


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PolusHipError("polus_amd ops need device (HBM) tensors; there is no CPU path")


_AUTO_SPLIT = {}


def gemm(a, b, out, *, a_layout=K_CONTIG, b_layout=K_CONTIG, M=None, N=None, K=None, alpha=1.0,
         bias=None, resid=None, aux=None, act=None, flags=0, split_k=1, drop_p=0.0, seed=0):
    """out[M,N] = epilogue(alpha * A_op . B_op); see include/polus_hip.h polus_gemm."""
    lib = _lib.load()
    _req_cuda(a, b, out, bias, resid, aux)
    dt = dtype_code(a.dtype)
    assert b.dtype == a.dtype and a.dim() == 2 and b.dim() == 2 and out.dim() == 2
    if a_layout == K_CONTIG:
        m_, k_ = a.shape
    else:
        k_, m_ = a.shape
    if b_layout == K_CONTIG:
        n_, kb_ = b.shape
    else:
        kb_, n_ = b.shape
    M = m_ if M is None else M
    N = n_ if N is None else N
    K = k_ if K is None else K
    assert kb_ == k_, f"contraction mismatch {k_} vs {kb_}"
    assert out.shape[0] >= M and out.shape[1] >= N
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    ws = None
    ws_bytes = 0
    if split_k == "auto":
        # the library's recommendation for a bf16 Dense GEMM (K-contiguous operands, bf16 C); 1 for everything else
        split_k = 1
        if (a.dtype == torch.bfloat16 and out.dtype == a.dtype and a_layout == K_CONTIG and b_layout == K_CONTIG
                and not (flags & GEMM_ACCUM_C)):
            key = (M, N, K)
            split_k = _AUTO_SPLIT.get(key)
            if split_k is None:
                split_k = _AUTO_SPLIT[key] = int(lib.polus_gemm_auto_split(M, N, K))
    if split_k > 1:
        ws_bytes = lib.polus_gemm_workspace_bytes(M, N, split_k)
        ws = workspace(a.device).get(ws_bytes)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    common = (dt, a_layout, b_layout, dtype_code(out.dtype),
              ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0),
              M, N, K, float(alpha), ptr(bias),
              ptr(resid), resid.stride(0) if resid is not None else 0,
              ptr(aux), aux.stride(0) if aux is not None else 0,
              ACT_CODES[act] if not isinstance(act, int) else act, flags, split_k, ptr(ws), ws_bytes)
    if drop_p > 0.0:
        check(lib.polus_gemm_dropout(*common, float(drop_p), int(seed) & 0xFFFFFFFF, _st()), "polus_gemm_dropout")
    else:
        check(lib.polus_gemm(*common, _st()), "polus_gemm")
    if prof is not None:
        e1.record()
        kind = "fwd" if (a_layout == K_CONTIG and b_layout == K_CONTIG) else ("dx" if a_layout == K_CONTIG else "dw")
        prof.append((kind, 2.0 * M * N * K, e0, e1))
    return out


def attention_fwd(qkv, mask, ctx, lse, B, S, n_heads, drop_p=0.0, seed=0):
    lib = _lib.load()
    _req_cuda(qkv, mask, ctx, lse)
    H = n_heads * 64
    assert qkv.shape == (B * S, 3 * H) and ctx.shape == (B * S, H) and qkv.is_contiguous() and ctx.is_contiguous()
    assert lse.dtype == torch.float32 and lse.numel() == B * n_heads * S
    assert mask is None or (mask.dtype == torch.int32 and mask.numel() == B * S and mask.is_contiguous())
    check(lib.polus_attention_fwd(dtype_code(qkv.dtype), ptr(qkv), ptr(mask), ptr(ctx), ptr(lse),
                                  B, S, n_heads, 64, float(drop_p), int(seed) & 0xFFFFFFFF, _st()), "polus_attention_fwd")


def attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, n_heads, drop_p=0.0, seed=0):
    lib = _lib.load()
    _req_cuda(qkv, mask, ctx, dctx, lse, dqkv)
    H = n_heads * 64
    assert qkv.shape == (B * S, 3 * H) and dqkv.shape == (B * S, 3 * H) and dctx.shape == (B * S, H)
    assert qkv.is_contiguous() and dqkv.is_contiguous() and dctx.is_contiguous() and ctx.is_contiguous()
    nb = lib.polus_attention_bwd_workspace_bytes(B, S, n_heads)
    ws = workspace(qkv.device).get(nb)
    check(lib.polus_attention_bwd(dtype_code(qkv.dtype), ptr(qkv), ptr(mask), ptr(ctx), ptr(dctx), ptr(lse),
                                  ptr(dqkv), B, S, n_heads, 64, float(drop_p), int(seed) & 0xFFFFFFFF, ptr(ws), nb, _st()),
          "polus_attention_bwd")


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps):
    lib = _lib.load()
    _req_cuda(x, gamma, beta, y, mean, rstd)
    rows, H = x.shape
    assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape
    check(lib.polus_layernorm_fwd(dtype_code(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd),
                                  rows, H, float(eps), _st()), "polus_layernorm_fwd")


def layernorm_bwd_partial_floats(rows, H):
    """f32 elements of a partial-sum buffer for layernorm_bwd(..., partials=...)."""
    return (_lib.load().polus_layernorm_bwd_workspace_bytes(rows, H) + 3) // 4


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias=None, accumulate=False, dx_masked=None,
                  drop_p=0.0, seed=0, partials=None):
    """`partials` (f32 tensor of layernorm_bwd_partial_floats elements, owned by the caller): stop after the main kernel and
    leave the per-workgroup sums there; dgamma / dbeta / dbias are then written by layernorm_bwd_finalize, which the caller
    queues on any stream ordered behind this call."""
    lib = _lib.load()
    _req_cuda(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias)
    rows, H = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous()
    nb = lib.polus_layernorm_bwd_workspace_bytes(rows, H)
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.numel() * 4 >= nb and partials.is_contiguous()
        check(lib.polus_layernorm_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                      None, None, ptr(dbias), int(accumulate), rows, H,
                                      ptr(dx_masked), float(drop_p), int(seed) & 0xFFFFFFFF, ptr(partials), partials.numel() * 4, _st()),
              "polus_layernorm_bwd")
        return
    ws = workspace(x.device).get(nb)
    check(lib.polus_layernorm_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                  ptr(dgamma), ptr(dbeta), ptr(dbias), int(accumulate), rows, H,
                                  ptr(dx_masked), float(drop_p), int(seed) & 0xFFFFFFFF, ptr(ws), nb, _st()), "polus_layernorm_bwd")


def layernorm_bwd_finalize(partials, rows, H, dgamma, dbeta, dbias=None, accumulate=False):
    _req_cuda(partials, dgamma, dbeta, dbias)
    check(_lib.load().polus_layernorm_bwd_finalize(ptr(partials), partials.numel() * 4, rows, H, ptr(dgamma), ptr(dbeta), ptr(dbias),
                                                   int(accumulate), _st()), "polus_layernorm_bwd_finalize")


def embed_ln_fwd(ids, type_ids, word, pos, typ, gamma, beta, y, mean, rstd, eps, drop_p=0.0, seed=0):
    lib = _lib.load()
    _req_cuda(ids, type_ids, word, pos, typ, gamma, beta, y, mean, rstd)
    B, S = ids.shape
    V, H = word.shape
    assert ids.dtype == torch.int32 and ids.is_contiguous()
    assert type_ids is None or (type_ids.dtype == torch.int32 and type_ids.is_contiguous())
    check(lib.polus_embed_ln_fwd(dtype_code(y.dtype), ptr(ids), ptr(type_ids), ptr(word), ptr(pos), ptr(typ),
                                 ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd),
                                 B, S, H, V, pos.shape[0], typ.shape[0], float(eps), float(drop_p), int(seed) & 0xFFFFFFFF,
                                 _st()), "polus_embed_ln_fwd")


def embed_ln_bwd(dy, ids, type_ids, word, pos, typ, gamma, mean, rstd, gword, gpos, gtyp, ggamma, gbeta,
                 accumulate=False, deterministic=False, drop_p=0.0, seed=0):
    lib = _lib.load()
    _req_cuda(dy, ids, word, gword)
    B, S = ids.shape
    V, H = word.shape
    nb = lib.polus_embed_bwd_workspace_bytes(B, S, H)
    ws = workspace(dy.device).get(nb)
    check(lib.polus_embed_ln_bwd(dtype_code(dy.dtype), ptr(dy), ptr(ids), ptr(type_ids), ptr(word), ptr(pos), ptr(typ),
                                 ptr(gamma), ptr(mean), ptr(rstd), ptr(gword), ptr(gpos), ptr(gtyp), ptr(ggamma),
                                 ptr(gbeta), int(accumulate), int(deterministic),
                                 B, S, H, V, pos.shape[0], typ.shape[0], float(drop_p), int(seed) & 0xFFFFFFFF,
                                 ptr(ws), nb, _st()), "polus_embed_ln_bwd")


def colsum(x, out, accumulate=False, rows=None, cols=None):
    lib = _lib.load()
    _req_cuda(x, out)
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    nb = lib.polus_colsum_workspace_bytes(rows, cols)
    ws = workspace(x.device).get(nb)
    check(lib.polus_colsum(dtype_code(x.dtype), ptr(x), x.stride(0), rows, cols, ptr(out), int(accumulate),
                           ptr(ws), nb, _st()), "polus_colsum")


def softmax_xent(logits, labels, loss, dlogits, class_weights=None, rows=None, C=None):
    lib = _lib.load()
    _req_cuda(logits, labels, loss, dlogits, class_weights)
    rows = logits.shape[0] if rows is None else rows
    C = logits.shape[1] if C is None else C
    assert logits.dtype == torch.float32 and labels.dtype == torch.int32 and loss.dtype == torch.float32
    nb = lib.polus_loss_workspace_bytes(rows)
    ws = workspace(logits.device).get(nb)
    check(lib.polus_softmax_xent(dtype_code(dlogits.dtype), ptr(logits), logits.stride(0), ptr(labels),
                                 ptr(class_weights), ptr(loss), ptr(dlogits), dlogits.stride(0), rows, C,
                                 ptr(ws), nb, _st()), "polus_softmax_xent")


def sigmoid_xent(logits, y_true, class_weights, negative_weight, loss, dlogits):
    lib = _lib.load()
    _req_cuda(logits, y_true, class_weights, loss, dlogits)
    rows, C = logits.shape
    nb = lib.polus_loss_workspace_bytes(rows)
    ws = workspace(logits.device).get(nb)
    check(lib.polus_sigmoid_xent(dtype_code(dlogits.dtype), ptr(logits), logits.stride(0), ptr(y_true), y_true.stride(0),
                                 ptr(class_weights), float(negative_weight), ptr(loss), ptr(dlogits), dlogits.stride(0),
                                 rows, C, ptr(ws), nb, _st()), "polus_sigmoid_xent")


def crf_nll(potentials, tags, lengths, trans, sample_w, loss, dpot, dtrans, accumulate=False):
    lib = _lib.load()
    _req_cuda(potentials, tags, lengths, trans, sample_w, loss, dpot, dtrans)
    B, S, C = potentials.shape
    assert potentials.dtype == torch.float32 and potentials.is_contiguous() and dpot.is_contiguous()
    nb = lib.polus_crf_workspace_bytes(B, S, C)
    ws = workspace(potentials.device).get(nb)
    check(lib.polus_crf_nll(dtype_code(dpot.dtype), ptr(potentials), ptr(tags), ptr(lengths), ptr(trans), ptr(sample_w),
                            ptr(loss), ptr(dpot), ptr(dtrans), int(accumulate), B, S, C, ptr(ws), nb, _st()),
          "polus_crf_nll")


def crf_viterbi(potentials, lengths, trans, out_tags):
    lib = _lib.load()
    _req_cuda(potentials, lengths, trans, out_tags)
    B, S, C = potentials.shape
    nb = lib.polus_crf_workspace_bytes(B, S, C)
    ws = workspace(potentials.device).get(nb)
    check(lib.polus_crf_viterbi(ptr(potentials), ptr(lengths), ptr(trans), ptr(out_tags), B, S, C, ptr(ws), nb, _st()),
          "polus_crf_viterbi")


def argmax(x, out, rows=None, C=None):
    lib = _lib.load()
    _req_cuda(x, out)
    rows = x.shape[0] if rows is None else rows
    C = x.shape[1] if C is None else C
    check(lib.polus_argmax(ptr(x), x.stride(0), ptr(out), rows, C, _st()), "polus_argmax")


def confusion_matrix(row_idx, col_idx, cm, rejected=None):
    """cm[row_idx[i], col_idx[i]] += 1 (int32 [C, C] on the device); pairs with an index outside [0, C) are counted in
    `rejected` (int32 [1] on the device) instead."""
    _req_cuda(row_idx, col_idx, cm)
    assert row_idx.dtype == torch.int32 and col_idx.dtype == torch.int32 and cm.dtype == torch.int32
    assert row_idx.numel() == col_idx.numel() and row_idx.is_contiguous() and col_idx.is_contiguous() and cm.is_contiguous()
    check(_lib.load().polus_confusion_matrix(ptr(row_idx), ptr(col_idx), row_idx.numel(), cm.shape[0], ptr(cm),
                                             ptr(rejected) if rejected is not None else None, _st()),
          "polus_confusion_matrix")


def adam_step(p, g, m, v, shadow, seg, n_seg, lr, lr_t, beta1, beta2, eps, weight_decay, grad_scale=1.0,
              clip_scale=None):
    lib = _lib.load()
    _req_cuda(p, g, m, v, shadow, seg, clip_scale)
    check(lib.polus_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), ptr(seg), n_seg, p.numel(),
                              float(lr), float(lr_t), float(beta1), float(beta2), float(eps), float(weight_decay),
                              float(grad_scale), ptr(clip_scale), _st()), "polus_adam_step")


def sqnorm(g, out):
    lib = _lib.load()
    _req_cuda(g, out)
    nb = lib.polus_sqnorm_workspace_bytes(g.numel())
    ws = workspace(g.device).get(nb)
    check(lib.polus_sqnorm(ptr(g), g.numel(), ptr(out), ptr(ws), nb, _st()), "polus_sqnorm")


def sqnorm_segments(g, seg, n_seg, out):
    """sum of g^2 over the windows of the Adam segment table `seg` (int64 [n_seg, 3]) -> out[0]."""
    lib = _lib.load()
    _req_cuda(g, seg, out)
    ws = workspace(g.device).get(4096)
    check(lib.polus_sqnorm_segments(ptr(g), ptr(seg), int(n_seg), ptr(out), ptr(ws), 4096, _st()), "polus_sqnorm_segments")


def clip_scale(sq, grad_scale, clip_norm, out):
    """out = clip_norm / max(sqrt(sum(sq)) * |grad_scale|, clip_norm); sq holds one partial sum per arena."""
    check(_lib.load().polus_clip_scale(ptr(sq), sq.numel(), float(grad_scale), float(clip_norm), ptr(out), _st()), "polus_clip_scale")


def cast(src, dst):
    _req_cuda(src, dst)
    assert src.numel() == dst.numel()
    check(_lib.load().polus_cast(dtype_code(src.dtype), ptr(src), dtype_code(dst.dtype), ptr(dst), src.numel(), _st()),
          "polus_cast")


def scale_(x, a):
    _req_cuda(x)
    check(_lib.load().polus_scale(ptr(x), float(a), x.numel(), _st()), "polus_scale")


def act_bwd(dy, u, du, act):
    _req_cuda(dy, u, du)
    assert dy.numel() == u.numel() == du.numel() and dy.is_contiguous() and u.is_contiguous() and du.is_contiguous()
    check(_lib.load().polus_act_bwd(dtype_code(dy.dtype), ptr(dy), ptr(u), ptr(du), dy.numel(),
                                    ACT_CODES[act] if not isinstance(act, int) else act, _st()), "polus_act_bwd")


def transpose_bf16(src, dst):
    _req_cuda(src, dst)
    R, C = src.shape
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16 and dst.numel() == src.numel()
    assert src.is_contiguous() and dst.is_contiguous()
    check(_lib.load().polus_transpose_bf16(ptr(src), ptr(dst), R, C, _st()), "polus_transpose_bf16")


def dense_bwd_params_grouped(problems, accumulate=False, split_k=1):
    """dW (+ db) of several Dense layers in one launch.  problems: [(dy [T,n_out], x [T,n_in], dw f32 [n_out,n_in],
    db f32 [n_out] or None), ...] with one common T."""
    lib = _lib.load()
    T = problems[0][0].shape[0]
    arr = (_lib.DwProblem * len(problems))()
    for k, (dy, x, dw, db) in enumerate(problems):
        _req_cuda(dy, x, dw, db)
        assert dy.dtype == x.dtype and dw.dtype == torch.float32 and dy.shape[0] == T and x.shape[0] == T
        assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
        n_out, n_in = dy.shape[1], x.shape[1]
        assert tuple(dw.shape) == (n_out, n_in) and (db is None or (db.dtype == torch.float32 and db.numel() == n_out))
        arr[k] = _lib.DwProblem(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(dw), dw.stride(0), ptr(db), n_out, n_in)
    dt = dtype_code(problems[0][0].dtype)
    nb = lib.polus_dense_bwd_params_grouped_workspace_bytes(len(problems), arr, T, int(split_k))
    ws = workspace(problems[0][0].device).get(nb)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.polus_dense_bwd_params_grouped(dt, len(problems), arr, T, 1 if accumulate else 0, int(split_k),
                                             ptr(ws), ws.numel(), _st()), "polus_dense_bwd_params_grouped")
    if prof is not None:
        e1.record()
        prof.append(("dw", sum(2.0 * T * p[0].shape[1] * p[1].shape[1] for p in problems), e0, e1))


def dense_thin_supported(dtype, H, C):
    """A Dense with at most 8 units takes the HBM-bound wave-per-row kernels instead of MFMA tiles (include/polus_hip.h)."""
    return bool(_lib.load().polus_dense_thin_supported(dtype_code(dtype), int(H), int(C)))


def dense_thin_fwd(x, w, bias, y):
    """y [rows, C] = x [rows, H] . w [C, H]^T + bias; y in f32 or x's dtype."""
    _req_cuda(x, w, bias, y)
    rows, H = x.shape
    C = w.shape[0]
    assert w.dtype == x.dtype and w.shape[1] == H and y.shape == (rows, C)
    assert x.stride(1) == 1 and w.stride(1) == 1 and y.stride(1) == 1
    check(_lib.load().polus_dense_thin_fwd(dtype_code(x.dtype), ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias),
                                           dtype_code(y.dtype), ptr(y), y.stride(0), rows, H, C, _st()), "polus_dense_thin_fwd")


def dense_thin_bwd(x, dy, w, dx, dw, db, accumulate=False):
    """dx [rows, H] = dy . w (dx None: skipped), dw [C, H] (+)= dy^T x, db [C] (+)= colsum(dy): one pass over x."""
    lib = _lib.load()
    _req_cuda(x, dy, w, dx, dw, db)
    rows, H = x.shape
    C = w.shape[0]
    assert dy.shape == (rows, C) and dy.stride(1) == 1 and dw.dtype == torch.float32 and tuple(dw.shape) == (C, H) and dw.stride(1) == 1
    assert dx is None or (dx.dtype == x.dtype and dx.shape == x.shape and dx.stride(1) == 1)
    nb = lib.polus_dense_thin_bwd_workspace_bytes(dtype_code(x.dtype), rows, H, C)
    ws = workspace(x.device).get(nb)
    check(lib.polus_dense_thin_bwd(dtype_code(x.dtype), ptr(x), x.stride(0), dtype_code(dy.dtype), ptr(dy), dy.stride(0),
                                   ptr(w), w.stride(0), ptr(dx), dx.stride(0) if dx is not None else 0, ptr(dw), dw.stride(0), ptr(db),
                                   rows, H, C, 1 if accumulate else 0, ptr(ws), ws.numel(), _st()), "polus_dense_thin_bwd")


def transpose_bf16_batched(src_base, dst_base, segs_dev, nseg, total_tiles):
    """All matrices of a flat bf16 arena into its transposed twin in one launch (segs: int64 [nseg, 4])."""
    _req_cuda(src_base, dst_base, segs_dev)
    assert src_base.dtype == torch.bfloat16 and dst_base.dtype == torch.bfloat16 and segs_dev.dtype == torch.int64
    check(_lib.load().polus_transpose_bf16_batched(ptr(src_base), ptr(dst_base), ptr(segs_dev), int(nseg), int(total_tiles), _st()),
          "polus_transpose_bf16_batched")


def dense_bwd_params(dy, x, dw, db, accumulate=False, split_k=1):
    """dW (+)= dY^T X and db (+)= colsum(dY) in one pass over dY."""
    lib = _lib.load()
    _req_cuda(dy, x, dw, db)
    T, n_out = dy.shape
    n_in = x.shape[1]
    assert x.shape[0] == T and dw.shape == (n_out, n_in) and dw.dtype == torch.float32
    assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
    nb = lib.polus_dense_bwd_params_workspace_bytes(T, n_out, n_in, split_k)
    ws = workspace(dy.device).get(nb)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.polus_dense_bwd_params(dtype_code(dy.dtype), ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(dw), dw.stride(0),
                                     ptr(db), T, n_out, n_in, int(accumulate), split_k, ptr(ws), nb, _st()),
          "polus_dense_bwd_params")
    if prof is not None:
        e1.record()
        prof.append(("dw", 2.0 * T * n_out * n_in, e0, e1))


def dropout(x, y, drop_p, seed):
    _req_cuda(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(_lib.load().polus_dropout(dtype_code(x.dtype), ptr(x), ptr(y), x.numel(), float(drop_p), int(seed) & 0xFFFFFFFF, _st()),
          "polus_dropout")


def dropout_mask(seed, drop_p, n, idx0=0, device=None):
    """uint8 keep-mask of elements idx0..idx0+n-1 for `seed` (reference for every dropout site)."""
    m = torch.empty(int(n), dtype=torch.uint8, device=device or "cuda")
    check(_lib.load().polus_dropout_mask(int(seed) & 0xFFFFFFFF, float(drop_p), int(idx0), int(n), ptr(m), _st()), "polus_dropout_mask")
    return m
