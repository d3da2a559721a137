"""NumPy restatement of the BERT encoder forward/backward the Polus hot path runs.

Test infrastructure only (see oracle/__init__.py).

Follows, as text:
  * polus/models.py:175-195  additive key mask ``(1 - m) * -10000`` shaped [B,1,1,S]
  * polus/models.py:201-216  layer loop; pooler_output = hidden[:, 0, :] (no dense/tanh)
  * SURVEY.md Appendix A     per-layer math of HF BERT (third-party, absent from the
                             reference tree): post-LN, exact-erf GELU, LN eps 1e-12
Weights use the PyTorch ``[out, in]`` layout (the reference loads them with
``from_pt=True``, polus/models.py:229); Q/K/V are fused into one ``[3H, H]`` matrix
whose row blocks are query / key / value.
"""
import math

import numpy as np

try:  # scipy is in the image; keep a pure-numpy erf for safety
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

MASK_VALUE = -10000.0  # polus/models.py:192
LN_EPS = 1e-12


class BertConfig:
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=LN_EPS):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps


# ----------------------------------------------------------------------------- primitives

def gelu(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    cdf = 0.5 * (1.0 + _erf(x / math.sqrt(2.0)))
    pdf = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    return cdf + x * pdf


def swish(x):
    return x / (1.0 + np.exp(-x))


def swish_grad(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return s * (1.0 + x * (1.0 - s))


def layer_norm_fwd(x, g, b, eps=LN_EPS):
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)  # biased
    rstd = 1.0 / np.sqrt(var + eps)
    y = (x - mean) * rstd * g + b
    return y, mean[..., 0], rstd[..., 0]


def layer_norm_bwd(dy, x, g, mean, rstd):
    """Returns dx, dg, db for y = LN(x)."""
    xhat = (x - mean[..., None]) * rstd[..., None]
    dxhat = dy * g
    H = x.shape[-1]
    dx = (dxhat - dxhat.mean(-1, keepdims=True)
          - xhat * (dxhat * xhat).mean(-1, keepdims=True)) * rstd[..., None]
    red = tuple(range(x.ndim - 1))
    dg = (dy * xhat).sum(red)
    db = dy.sum(red)
    return dx, dg, db


def linear_fwd(x, w, b=None):
    """x [..., in] · w[out, in]^T + b."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def linear_bwd(dy, x, w):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    dw = dy2.T @ x2
    db = dy2.sum(0)
    dx = dy @ w
    return dx, dw, db


def additive_mask(attention_mask, dtype=np.float32):
    """polus/models.py:175-195 — [B,S] {0,1} -> [B,1,1,S] {0,-10000}."""
    m = attention_mask.astype(dtype)
    return ((1.0 - m) * MASK_VALUE).reshape(m.shape[0], 1, 1, m.shape[1]).astype(dtype)


def attention_fwd(qkv, add_mask, n_heads, keep_scale=None):
    """qkv [B,S,3H] fused; add_mask [B,1,1,S]. Returns ctx [B,S,H], probs [B,A,S,S].
    keep_scale [B,A,S,S] = mask/(1-p): HF attention_probs_dropout, applied after the softmax."""
    B, S, H3 = qkv.shape
    H = H3 // 3
    d = H // n_heads
    q = qkv[..., :H].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    k = qkv[..., H:2 * H].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    v = qkv[..., 2 * H:].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    scores = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(d) + add_mask
    scores = scores - scores.max(-1, keepdims=True)
    e = np.exp(scores)
    probs = e / e.sum(-1, keepdims=True)
    pd = probs if keep_scale is None else probs * keep_scale
    ctx = (pd @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
    return ctx, probs


def attention_bwd(dctx, qkv, probs, n_heads, keep_scale=None):
    B, S, H3 = qkv.shape
    H = H3 // 3
    d = H // n_heads
    q = qkv[..., :H].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    k = qkv[..., H:2 * H].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    v = qkv[..., 2 * H:].reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    do = dctx.reshape(B, S, n_heads, d).transpose(0, 2, 1, 3)
    pd = probs if keep_scale is None else probs * keep_scale
    dv = pd.transpose(0, 1, 3, 2) @ do
    dp = do @ v.transpose(0, 1, 3, 2)
    if keep_scale is not None:
        dp = dp * keep_scale
    ds = probs * (dp - (dp * probs).sum(-1, keepdims=True))
    ds = ds / math.sqrt(d)
    dq = ds @ k
    dk = ds.transpose(0, 1, 3, 2) @ q
    def merge(t):
        return t.transpose(0, 2, 1, 3).reshape(B, S, H)
    return np.concatenate([merge(dq), merge(dk), merge(dv)], axis=-1)


# ----------------------------------------------------------------------------- parameters

def layer_param_shapes(cfg):
    H, I = cfg.hidden_size, cfg.intermediate_size
    return [("qkv.w", (3 * H, H)), ("qkv.b", (3 * H,)),
            ("out.w", (H, H)), ("out.b", (H,)),
            ("ln1.g", (H,)), ("ln1.b", (H,)),
            ("ffn1.w", (I, H)), ("ffn1.b", (I,)),
            ("ffn2.w", (H, I)), ("ffn2.b", (H,)),
            ("ln2.g", (H,)), ("ln2.b", (H,))]


def param_shapes(cfg, with_embeddings=True, layers=None):
    """Ordered (name, shape) list — the order of the flat parameter arena."""
    H = cfg.hidden_size
    out = []
    if with_embeddings:
        out += [("emb.word", (cfg.vocab_size, H)),
                ("emb.pos", (cfg.max_position_embeddings, H)),
                ("emb.type", (cfg.type_vocab_size, H)),
                ("emb.ln.g", (H,)), ("emb.ln.b", (H,))]
    for i in (range(cfg.num_hidden_layers) if layers is None else layers):
        out += [(f"layer{i}.{n}", s) for n, s in layer_param_shapes(cfg)]
    return out


def init_params(cfg, seed=1234, std=0.02, dtype=np.float32, with_embeddings=True):
    """N(0, 0.02) truncated at 2 sigma for matrices, LN gamma=1 beta=0, biases 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = {}
    for name, shape in param_shapes(cfg, with_embeddings):
        if name.endswith(".g"):
            p[name] = np.ones(shape, dtype)
        elif name.endswith(".b"):
            p[name] = np.zeros(shape, dtype)
        else:
            w = rng.standard_normal(shape)
            w = np.clip(w, -2.0, 2.0) * std
            p[name] = w.astype(dtype)
    return p


# ----------------------------------------------------------------------------- model

def embeddings_fwd(p, cfg, input_ids, token_type_ids=None, keep_scale=None):
    B, S = input_ids.shape
    if token_type_ids is None:
        token_type_ids = np.zeros_like(input_ids)
    e = p["emb.word"][input_ids] + p["emb.type"][token_type_ids] + p["emb.pos"][:S][None]
    y, mean, rstd = layer_norm_fwd(e, p["emb.ln.g"], p["emb.ln.b"], cfg.layer_norm_eps)
    if keep_scale is not None:
        y = y * keep_scale
    return y, (e, mean, rstd, input_ids, token_type_ids, keep_scale)


def embeddings_bwd(dy, p, cfg, cache):
    e, mean, rstd, input_ids, token_type_ids, keep_scale = cache
    if keep_scale is not None:
        dy = dy * keep_scale
    de, dg, db = layer_norm_bwd(dy, e, p["emb.ln.g"], mean, rstd)
    g = {"emb.ln.g": dg, "emb.ln.b": db}
    gw = np.zeros_like(p["emb.word"])
    np.add.at(gw, input_ids.reshape(-1), de.reshape(-1, de.shape[-1]))
    gt = np.zeros_like(p["emb.type"])
    np.add.at(gt, token_type_ids.reshape(-1), de.reshape(-1, de.shape[-1]))
    gp = np.zeros_like(p["emb.pos"])
    gp[:de.shape[1]] = de.sum(0)
    g["emb.word"], g["emb.type"], g["emb.pos"] = gw, gt, gp
    return g


def layer_fwd(p, cfg, i, x, add_mask, drop=None):
    """drop = dict(att=[B,A,S,S], h1=[B,S,H], h2=[B,S,H]) of keep-scales mask/(1-p), or None."""
    pre = f"layer{i}."
    d = drop or {}
    qkv = linear_fwd(x, p[pre + "qkv.w"], p[pre + "qkv.b"])
    ctx, probs = attention_fwd(qkv, add_mask, cfg.num_attention_heads, d.get("att"))
    o1 = linear_fwd(ctx, p[pre + "out.w"], p[pre + "out.b"])
    z1 = (o1 if d.get("h1") is None else o1 * d["h1"]) + x
    a1, m1, r1 = layer_norm_fwd(z1, p[pre + "ln1.g"], p[pre + "ln1.b"], cfg.layer_norm_eps)
    u = linear_fwd(a1, p[pre + "ffn1.w"], p[pre + "ffn1.b"])
    f = gelu(u)
    o2 = linear_fwd(f, p[pre + "ffn2.w"], p[pre + "ffn2.b"])
    z2 = (o2 if d.get("h2") is None else o2 * d["h2"]) + a1
    y, m2, r2 = layer_norm_fwd(z2, p[pre + "ln2.g"], p[pre + "ln2.b"], cfg.layer_norm_eps)
    cache = dict(x=x, qkv=qkv, probs=probs, ctx=ctx, z1=z1, m1=m1, r1=r1, a1=a1,
                 u=u, f=f, z2=z2, m2=m2, r2=r2, drop=d)
    return y, cache


def layer_bwd(dy, p, cfg, i, c):
    pre = f"layer{i}."
    g = {}
    d = c.get("drop") or {}
    dz2, g[pre + "ln2.g"], g[pre + "ln2.b"] = layer_norm_bwd(dy, c["z2"], p[pre + "ln2.g"], c["m2"], c["r2"])
    dz2d = dz2 if d.get("h2") is None else dz2 * d["h2"]
    df, g[pre + "ffn2.w"], g[pre + "ffn2.b"] = linear_bwd(dz2d, c["f"], p[pre + "ffn2.w"])
    du = df * gelu_grad(c["u"])
    da1, g[pre + "ffn1.w"], g[pre + "ffn1.b"] = linear_bwd(du, c["a1"], p[pre + "ffn1.w"])
    da1 = da1 + dz2
    dz1, g[pre + "ln1.g"], g[pre + "ln1.b"] = layer_norm_bwd(da1, c["z1"], p[pre + "ln1.g"], c["m1"], c["r1"])
    dz1d = dz1 if d.get("h1") is None else dz1 * d["h1"]
    dctx, g[pre + "out.w"], g[pre + "out.b"] = linear_bwd(dz1d, c["ctx"], p[pre + "out.w"])
    dqkv = attention_bwd(dctx, c["qkv"], c["probs"], cfg.num_attention_heads, d.get("att"))
    dx, g[pre + "qkv.w"], g[pre + "qkv.b"] = linear_bwd(dqkv, c["x"], p[pre + "qkv.w"])
    dx = dx + dz1
    return dx, g


def encoder_fwd(p, cfg, hidden, attention_mask, layers=None, drop=None):
    """The TFBertSplited.call restatement (polus/models.py:197-216): runs `layers`
    (default: all) over `hidden` with the -10000 additive mask; returns
    (last_hidden_state, pooler_output=hidden[:,0,:], caches)."""
    add_mask = additive_mask(attention_mask, hidden.dtype)
    caches = []
    for i in (range(cfg.num_hidden_layers) if layers is None else layers):
        hidden, c = layer_fwd(p, cfg, i, hidden, add_mask, None if drop is None else drop["layers"][i])
        caches.append((i, c))
    return hidden, hidden[:, 0, :], caches


def encoder_bwd(dhidden, p, cfg, caches):
    grads = {}
    for i, c in reversed(caches):
        dhidden, g = layer_bwd(dhidden, p, cfg, i, c)
        grads.update(g)
    return dhidden, grads


def bert_fwd(p, cfg, input_ids, attention_mask, token_type_ids=None, drop=None):
    """drop = dict(emb=[B,S,H], layers=[dict(att,h1,h2), ...]) of keep-scales, or None (dropout 0)."""
    emb, ecache = embeddings_fwd(p, cfg, input_ids, token_type_ids, None if drop is None else drop["emb"])
    last, pooled, caches = encoder_fwd(p, cfg, emb, attention_mask, drop=drop)
    return last, pooled, (ecache, caches)


def bert_bwd(dlast, p, cfg, cache):
    ecache, caches = cache
    demb, grads = encoder_bwd(dlast, p, cfg, caches)
    grads.update(embeddings_bwd(demb, p, cfg, ecache))
    return grads


# ----------------------------------------------------------------------------- token classification

def token_classifier_fwd(p, cfg, head_w, head_b, input_ids, attention_mask, labels,
                         token_type_ids=None, drop=None):
    """BERT + Dense(H->C) + sparse softmax CE (mean over every position), the
    ClassifierTrainer step shape of polus/training.py:366-397 with the Keras loss
    of tutorials/classifier_example.py:55."""
    from .losses import sparse_softmax_xent_fwd
    last, _, cache = bert_fwd(p, cfg, input_ids, attention_mask, token_type_ids, drop)
    logits = linear_fwd(last, head_w, head_b)
    loss, dlogits = sparse_softmax_xent_fwd(logits, labels)
    return loss, logits, (last, dlogits, cache)


def token_classifier_bwd(p, cfg, head_w, cache_all):
    last, dlogits, cache = cache_all
    dlast, dw, db = linear_bwd(dlogits, last, head_w)
    grads = bert_bwd(dlast, p, cfg, cache)
    grads["head.w"], grads["head.b"] = dw, db
    return grads


def golden_setup(cfg, n_classes, dtype=np.float64):
    """Seeded parameters of the committed golden cases (tests/golden/make_golden.py):
    init_params(seed 1234), then LN gains / all biases perturbed (seed 77) so that they
    carry signal; head = Dense(H -> C)."""
    params = init_params(cfg, seed=1234, dtype=np.float64)
    rng = np.random.Generator(np.random.PCG64(77))
    head_w = np.clip(rng.standard_normal((n_classes, cfg.hidden_size)), -2, 2) * 0.02
    head_b = np.zeros(n_classes)
    for k in params:
        if k.endswith(".g"):
            params[k] = params[k] + 0.1 * rng.standard_normal(params[k].shape)
        elif k.endswith(".b"):
            params[k] = 0.05 * rng.standard_normal(params[k].shape)
    params = {k: v.astype(dtype) for k, v in params.items()}
    return params, head_w.astype(dtype), head_b.astype(dtype)


# ----------------------------------------------------------------------------- HF pooler

def pooler_fwd(last_hidden, w, b):
    """HF BertPooler (transformers modeling_bert.py:451-463; what the reference reads as
    `pooler_output` from an UNSPLIT TFBertModel, polus/data.py:526-543): tanh(h[:, 0] W^T + b).
    TFBertSplited returns the raw slice instead (polus/models.py:215-216, bert_fwd's second output)."""
    cls = last_hidden[:, 0, :]
    pooled = np.tanh(cls @ w.T + b)
    return pooled, (cls, pooled)


def pooler_bwd(dpooled, w, cache, seq_len):
    """-> (d last_hidden [B,S,H], dW, db)."""
    cls, pooled = cache
    du = dpooled * (1.0 - pooled * pooled)
    dlast = np.zeros((cls.shape[0], seq_len, cls.shape[1]), dtype=cls.dtype)
    dlast[:, 0, :] = du @ w
    return dlast, du.T @ cls, du.sum(0)
