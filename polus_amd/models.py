"""Models of the Polus hot path on MI355X: the BERT encoder with explicit forward/backward
over HIP kernels, the Keras-like Sequential wrappers, and the reference's model helpers.

Mirrors polus/models.py: PolusModel / SavableModel / PolusClassifier (:85-154),
TFBertSplited (:164-216), split_bert_model (:242-295).  The HuggingFace TFBertModel the
reference loads (polus/models.py:225-229) is replaced by `BertModel` below: same
architecture, weights in PyTorch [out, in] layout with Q/K/V fused
into one [3H, H] matrix.
"""
import json
import os
import sys
from functools import wraps

import numpy as np
import torch

from . import _lib, ops
from .layers import CRF, Dense, Dropout, Flatten, Layer, dense_bwd_params_group, gemm_dx
from .tensor import ParamArena, to_device, device
from .utils import complex_json_deserializer, complex_json_serializer, flatten_dict, merge_dicts

COMPUTE_DTYPES = {"f32": torch.float32, "float32": torch.float32, torch.float32: torch.float32,
                  "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, torch.bfloat16: torch.bfloat16}


def load_model(file_name_w_ext, change_config={}, external_module=None):
    """polus/models.py:18-50: rebuild a saved model from `<name>.cfg` (JSON written by SavableModel.save:
    the keyword arguments of its @from_config builder + `func_name`), replay the build-time sample of
    `<name>.init` if present, then load the weights (`weight0..N` in get_weights() order; `<name>.npz`
    here, `<name>.h5` in the reference -- h5py is not in this image).  The builder is looked up in
    `external_module` or in this module."""
    file_name = os.path.splitext(file_name_w_ext)[0]
    with open(file_name_w_ext, "r") as f:
        cfg = complex_json_deserializer(json.load(f))
    cfg["model"] = merge_dicts(cfg.get("model", {}), change_config)
    module = external_module if external_module is not None else sys.modules[__name__]
    model = getattr(module, cfg["func_name"])(**cfg)
    if os.path.exists(file_name + ".init.npz"):
        z = np.load(file_name + ".init.npz", allow_pickle=False)
        spec = json.loads(str(z["__spec__"]))
        args = [z[k] for k in spec["args"]]
        kwargs = {k: z[v] for k, v in spec["kwargs"].items()}
        model.init_from_data(*args, **kwargs)
    from .checkpoint import load_weights
    load_weights(model, file_name)
    return model


def resolve_activation(activation_name):
    """polus/models.py:53-57 (tfa.activations.mish has no kernel here: the name passes through and the
    Dense layer rejects what it cannot run)."""
    return activation_name


def from_config(func):
    """polus/models.py:60-82: the builder is called with nested keyword dicts (at least `model={...}`);
    it receives them flattened, and the model remembers the nested form + the builder's name as
    `savable_config`, which is what `.cfg` holds and `load_model` replays.  Flat keyword arguments are
    accepted too and filed under `model`."""
    @wraps(func)
    def function_wrapper(**kwargs):
        if "model" not in kwargs:
            kwargs = {"model": {k: v for k, v in kwargs.items() if k != "func_name"}}
        kwargs = {k: v for k, v in kwargs.items() if k != "func_name"}
        if "activation" in kwargs["model"]:
            _activation = kwargs["model"]["activation"]
            kwargs["model"]["activation"] = resolve_activation(_activation)
        model = func(**flatten_dict(kwargs))
        kwargs["func_name"] = func.__name__
        if "activation" in kwargs["model"]:
            kwargs["model"]["activation"] = _activation
        model._name = func.__name__
        model.savable_config = kwargs
        return model
    return function_wrapper


class BertConfig:
    """Hyper-parameters of HF BertConfig that the encoder math depends on."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, _name_or_path=""):
        if hidden_size != num_attention_heads * 64:
            raise ValueError("the fused attention kernel is built for head_dim 64 (BERT-base/large, BioBERT, PubMedBERT)")
        if not (0.0 <= hidden_dropout_prob < 1.0 and 0.0 <= attention_probs_dropout_prob < 1.0):
            raise ValueError("dropout probabilities must be in [0, 1)")
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size, self.layer_norm_eps = type_vocab_size, layer_norm_eps
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self._name_or_path = _name_or_path

    @classmethod
    def bert_base(cls, **kw):
        return cls(**kw)

    @classmethod
    def bert_large(cls, **kw):
        return cls(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, **kw)

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items()}


class BaseModelOutputWithPooling:
    """Stand-in for transformers' TFBaseModelOutputWithPooling (polus/models.py:215)."""

    def __init__(self, last_hidden_state, pooler_output):
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output

    def __getitem__(self, k):
        if isinstance(k, int):
            return (self.last_hidden_state, self.pooler_output)[k]
        return getattr(self, k)


_M32 = 0xFFFFFFFF


def dropout_seed_static(base, layer, site):
    """The step-independent part of a dropout site's seed (sites: 0 attention probs, 1 attention output,
    2 FFN output; layer -1 = embeddings)."""
    return (base * 0x9E3779B1 + ((layer + 1) * 8 + site) * 0xC2B2AE35) & _M32


def dropout_salt(step):
    """The per-step part: seed = mix(static + salt), mix(x) = (x ^ x >> 15) * 0x2C1B3C6D (uint32)."""
    return (step * 0x85EBCA6B) & _M32


def dropout_seed(base, step, layer, site):
    """32-bit seed of one dropout site of one step.  The kernels hash (seed, element index).  When a step is
    replayed from a HIP graph the kernels do the final mix themselves from the static part and a salt in
    device memory (include/polus_hip.h polus_set_dynamic_params): same masks either way."""
    x = (dropout_seed_static(base, layer, site) + dropout_salt(step)) & _M32
    x ^= x >> 15
    return (x * 0x2C1B3C6D) & _M32


def _trunc_normal(rng, shape, std=0.02):
    return (np.clip(rng.standard_normal(shape), -2.0, 2.0) * std).astype(np.float32)


# ------------------------------------------------------------------------------------ BERT pieces
class BertLayer:
    """One post-LN transformer block: 4 MFMA GEMMs + fused attention + 2 LayerNorms forward,
    8 GEMMs + 2 attention-backward kernels + 2 LN-backward backward."""

    def __init__(self, cfg, arena, index, rng):
        H, I = cfg.hidden_size, cfg.intermediate_size
        self.cfg, self.index = cfg, index
        p = f"layer{index}."
        add = arena.add
        self.qkv_w = add(p + "qkv.w", (3 * H, H), _trunc_normal(rng, (3 * H, H)), matrix=True)
        self.qkv_b = add(p + "qkv.b", (3 * H,), np.zeros(3 * H, np.float32), decay=False)
        self.out_w = add(p + "out.w", (H, H), _trunc_normal(rng, (H, H)), matrix=True)
        self.out_b = add(p + "out.b", (H,), np.zeros(H, np.float32), decay=False)
        self.ln1_g = add(p + "ln1.g", (H,), np.ones(H, np.float32), decay=False)
        self.ln1_b = add(p + "ln1.b", (H,), np.zeros(H, np.float32), decay=False)
        self.ffn1_w = add(p + "ffn1.w", (I, H), _trunc_normal(rng, (I, H)), matrix=True)
        self.ffn1_b = add(p + "ffn1.b", (I,), np.zeros(I, np.float32), decay=False)
        self.ffn2_w = add(p + "ffn2.w", (H, I), _trunc_normal(rng, (H, I)), matrix=True)
        self.ffn2_b = add(p + "ffn2.b", (H,), np.zeros(H, np.float32), decay=False)
        self.ln2_g = add(p + "ln2.g", (H,), np.ones(H, np.float32), decay=False)
        self.ln2_b = add(p + "ln2.b", (H,), np.zeros(H, np.float32), decay=False)
        self._bufs = {}

    def variables(self):
        return [self.qkv_w, self.qkv_b, self.out_w, self.out_b, self.ln1_g, self.ln1_b,
                self.ffn1_w, self.ffn1_b, self.ffn2_w, self.ffn2_b, self.ln2_g, self.ln2_b]

    def _buf(self, key, shape, dtype, dev):
        t = self._bufs.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype:
            t = self._bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=dev)
        return t

    def forward(self, x, mask, B, S, p_hid=0.0, p_att=0.0, seeds=(0, 0, 0), for_backward=True):
        """`for_backward=False` (inference, the frozen encoders of the dual-encoder trainer): the GELU pre-activation is
        not written -- half of FFN1's output bytes, a launch that is bound by exactly those."""
        cfg = self.cfg
        H, I, A = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
        T, dt, dev = B * S, x.dtype, x.device
        b = lambda k, shape, d=dt: self._buf(k, shape, d, dev)
        qkv, ctx, lse = b("qkv", (T, 3 * H)), b("ctx", (T, H)), b("lse", (B, A, S), torch.float32)
        z1, a1 = b("z1", (T, H)), b("a1", (T, H))
        m1, r1 = b("m1", (T,), torch.float32), b("r1", (T,), torch.float32)
        u, f = (b("u", (T, I)) if for_backward else None), b("f", (T, I))
        z2, y = b("z2", (T, H)), b("y", (T, H))
        m2, r2 = b("m2", (T,), torch.float32), b("r2", (T,), torch.float32)
        ops.gemm(x, self.qkv_w.compute, qkv, bias=self.qkv_b.value, split_k="auto")
        ops.attention_fwd(qkv, mask, ctx, lse, B, S, A, drop_p=p_att, seed=seeds[0])
        ops.gemm(ctx, self.out_w.compute, z1, bias=self.out_b.value, resid=x, drop_p=p_hid, seed=seeds[1], split_k="auto")
        ops.layernorm_fwd(z1, self.ln1_g.value, self.ln1_b.value, a1, m1, r1, cfg.layer_norm_eps)
        ops.gemm(a1, self.ffn1_w.compute, f, bias=self.ffn1_b.value, aux=u, act="gelu", flags=ops.GEMM_ACT_FWD, split_k="auto")
        ops.gemm(f, self.ffn2_w.compute, z2, bias=self.ffn2_b.value, resid=a1, drop_p=p_hid, seed=seeds[2], split_k="auto")
        ops.layernorm_fwd(z2, self.ln2_g.value, self.ln2_b.value, y, m2, r2, cfg.layer_norm_eps)
        self._stash = (x, mask, B, S, p_hid, p_att, seeds) if for_backward else None
        return y

    def backward(self, dy, scratch, accumulate=False, side=None):
        """dy [T,H] -> dx [T,H]; parameter gradients land in the arena (overwritten, or added
        to when `accumulate`).  `scratch(key, shape)` hands out buffers shared by all layers.
        The four weight gradients are ONE grouped launch (dense_bwd_params_grouped), queued once
        dqkv exists.  `side`: a second HIP stream for it: it then runs beside the last dX GEMM of
        this layer and the start of the next one.  In that mode the dY buffers it reads are
        double-buffered by layer parity and `self.dw_done` (an event on `side`) tells the caller
        when the gradients are final and the buffers free (BertModel.backward waits on it two
        layers later)."""
        cfg = self.cfg
        H, I, A = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
        if self._stash is None:
            raise RuntimeError("BertLayer.backward: the last forward pass ran with training=False (nothing was kept for backward)")
        x, mask, B, S, p_hid, p_att, seeds = self._stash
        T = B * S
        bb = self._bufs
        par = "" if side is None else str(self.index & 1)
        dz2, du = scratch("dz2" + par, (T, H)), scratch("du" + par, (T, I))
        da1, dz1 = scratch("da1", (T, H)), scratch("dz1" + par, (T, H))
        dctx, dqkv = scratch("dctx", (T, H)), scratch("dqkv" + par, (T, 3 * H))
        dx = scratch("dx%d" % (self.index & 1), (T, H))
        # z2 = dropout(ffn2(f)) + a1: the residual path takes dz2, the Dense path dz2 * mask/(1-p)
        dz2m = scratch("dz2m" + par, (T, H)) if p_hid > 0 else None
        # With the side stream, the two small reductions that finish dgamma / dbeta / the Dense bias gradients of the layer's
        # LayerNorms leave the critical path: the main kernels keep their per-workgroup partial sums (buffers by layer
        # parity, like the dY tensors) and the finalisations are queued behind the weight-gradient launch on `side`.
        lnp = None
        if side is not None and os.environ.get("POLUS_LN_DEFER_FINALIZE", "1") != "0":
            nfl = ops.layernorm_bwd_partial_floats(T, H)
            lnp = [scratch("lnp2" + par, (nfl,), torch.float32), scratch("lnp1" + par, (nfl,), torch.float32)]
        ops.layernorm_bwd(dy, bb["z2"], self.ln2_g.value, bb["m2"], bb["r2"], dz2,
                          self.ln2_g.grad, self.ln2_b.grad, self.ffn2_b.grad, accumulate,
                          dx_masked=dz2m, drop_p=p_hid, seed=seeds[2], partials=lnp[0] if lnp else None)
        dz2d = dz2m if dz2m is not None else dz2
        gemm_dx(dz2d, self.ffn2_w, du, aux=bb["u"], act="gelu", flags=ops.GEMM_ACT_BWD)
        gemm_dx(du, self.ffn1_w, da1, resid=dz2)
        dz1m = scratch("dz1m" + par, (T, H)) if p_hid > 0 else None
        ops.layernorm_bwd(da1, bb["z1"], self.ln1_g.value, bb["m1"], bb["r1"], dz1,
                          self.ln1_g.grad, self.ln1_b.grad, self.out_b.grad, accumulate,
                          dx_masked=dz1m, drop_p=p_hid, seed=seeds[1], partials=lnp[1] if lnp else None)
        dz1d = dz1m if dz1m is not None else dz1
        gemm_dx(dz1d, self.out_w, dctx)
        ops.attention_bwd(bb["qkv"], mask, bb["ctx"], dctx, bb["lse"], dqkv, B, S, A, drop_p=p_att, seed=seeds[0])
        problems = [(dz2d, bb["f"], self.ffn2_w.grad, None), (du, bb["a1"], self.ffn1_w.grad, self.ffn1_b.grad),
                    (dz1d, bb["ctx"], self.out_w.grad, None), (dqkv, x, self.qkv_w.grad, self.qkv_b.grad)]
        if side is None:
            dense_bwd_params_group(problems, accumulate)
            self.dw_done = None
        else:
            main = torch.cuda.current_stream()
            if not hasattr(self, "_ev"):
                self._ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev[0].record(main)
            side.wait_event(self._ev[0])
            with _lib.stream_scope(side):
                dense_bwd_params_group(problems, accumulate)
                if lnp:
                    ops.layernorm_bwd_finalize(lnp[0], T, H, self.ln2_g.grad, self.ln2_b.grad, self.ffn2_b.grad, accumulate)
                    ops.layernorm_bwd_finalize(lnp[1], T, H, self.ln1_g.grad, self.ln1_b.grad, self.out_b.grad, accumulate)
                self._ev[1].record(side)
            self.dw_done = self._ev[1]
        gemm_dx(dqkv, self.qkv_w, dx, resid=dz1)
        return dx


class BertEmbeddings:
    def __init__(self, cfg, arena, rng):
        H = cfg.hidden_size
        self.cfg = cfg
        self.word = arena.add("emb.word", (cfg.vocab_size, H), _trunc_normal(rng, (cfg.vocab_size, H)))
        self.pos = arena.add("emb.pos", (cfg.max_position_embeddings, H), _trunc_normal(rng, (cfg.max_position_embeddings, H)))
        self.type = arena.add("emb.type", (cfg.type_vocab_size, H), _trunc_normal(rng, (cfg.type_vocab_size, H)))
        self.ln_g = arena.add("emb.ln.g", (H,), np.ones(H, np.float32), decay=False)
        self.ln_b = arena.add("emb.ln.b", (H,), np.zeros(H, np.float32), decay=False)
        self._bufs = {}

    def variables(self):
        return [self.word, self.pos, self.type, self.ln_g, self.ln_b]

    def forward(self, input_ids, token_type_ids, dtype, p_hid=0.0, seed=0):
        B, S = input_ids.shape
        dev = input_ids.device
        key = (B, S, dtype)
        if self._bufs.get("key") != key:
            self._bufs = {"key": key,
                          "y": torch.empty((B * S, self.cfg.hidden_size), dtype=dtype, device=dev),
                          "mean": torch.empty(B * S, dtype=torch.float32, device=dev),
                          "rstd": torch.empty(B * S, dtype=torch.float32, device=dev)}
        bb = self._bufs
        ops.embed_ln_fwd(input_ids, token_type_ids, self.word.value, self.pos.value, self.type.value,
                         self.ln_g.value, self.ln_b.value, bb["y"], bb["mean"], bb["rstd"], self.cfg.layer_norm_eps,
                         drop_p=p_hid, seed=seed)
        self._stash = (input_ids, token_type_ids, p_hid, seed)
        return bb["y"]

    def backward(self, dy, accumulate=False, deterministic=False):
        ids, tts, p_hid, seed = self._stash
        bb = self._bufs
        ops.embed_ln_bwd(dy, ids, tts, self.word.value, self.pos.value, self.type.value, self.ln_g.value,
                         bb["mean"], bb["rstd"], self.word.grad, self.pos.grad, self.type.grad,
                         self.ln_g.grad, self.ln_b.grad, accumulate=accumulate, deterministic=deterministic,
                         drop_p=p_hid, seed=seed)


# ------------------------------------------------------------------------------------ model wrappers
class PolusModel:
    """polus/models.py:85-105.  Callable like a Keras model: ``model(x, training=False)`` or
    ``model(**x, training=False)``; ``backward(dy)`` fills the gradient arena;
    ``trainable_weights`` lists the Variables."""

    def __init__(self, name=None):
        self._name = name or self.__class__.__name__.lower()
        self.grad_ready_hook = None   # called as hook(lo, hi, variables) when the gradients of `variables` (arena window [lo, hi)) are final
        self.deterministic = False
        self.savable_config = {}

    @property
    def name(self):
        return self._name

    def set_name(self, name):
        self._name = name

    def init_from_data(self, *args, **kwargs):
        self._init = (args, kwargs)
        return self(*args, **kwargs)

    @property
    def trainable_weights(self):
        return list(self.arena.vars)

    def __call__(self, *args, training=False, **kwargs):
        return self.call(*args, training=training, **kwargs)

    def get_weights(self):
        return [v.numpy() for v in self.trainable_weights]

    def set_weights(self, weights):
        for v, w in zip(self.trainable_weights, weights):
            v.assign(w)

    def _notify(self, variables):
        if self.grad_ready_hook is not None and variables:
            lo = min(v.offset for v in variables)
            hi = max(v.offset + v.size for v in variables)
            self.grad_ready_hook(lo, hi, variables)


def _to_numpy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class SavableModel(PolusModel):
    """polus/models.py:107-133: <name><ext>.cfg (JSON config) + weights.  h5py is not in the
    image: weights go to <name><ext>.npz with keys weight0..N in get_weights() order, the
    same order the reference's .h5 datasets use."""

    def save(self, base_path=os.path.join(".polus_cache", "saved_models"), extension=""):
        os.makedirs(base_path, exist_ok=True)
        path = os.path.join(base_path, self.name + extension)
        with open(path + ".cfg", "w") as f:
            json.dump(complex_json_serializer(self.savable_config), f)
        if hasattr(self, "_init"):
            # the reference pickles (args, kwargs) into <name>.init; arrays go to an .npz here (nothing to execute on load)
            args, kwargs = self._init
            arrs, spec = {}, {"args": [], "kwargs": {}}
            for i, a in enumerate(args):
                arrs[f"arg{i}"] = _to_numpy(a); spec["args"].append(f"arg{i}")
            for k, a in kwargs.items():
                if k == "training":
                    continue
                arrs[f"kw_{k}"] = _to_numpy(a); spec["kwargs"][k] = f"kw_{k}"
            np.savez(path + ".init.npz", __spec__=np.asarray(json.dumps(spec)), **arrs)
        w = self.get_weights()
        np.savez(path + ".npz", **{f"weight{i}": a for i, a in enumerate(w)})
        return path


class PolusClassifier(SavableModel):
    def inference(self, x):
        """polus/models.py:148-150: argmax over the last axis, int32."""
        logits = self(x, training=False)
        l2 = logits.reshape(-1, logits.shape[-1])
        if l2.dtype != torch.float32:
            l2 = l2.float()
        out = torch.empty(l2.shape[0], dtype=torch.int32, device=l2.device)
        ops.argmax(l2.contiguous(), out)
        return out.view(logits.shape[:-1])


class Sequential(PolusModel):
    """tf.keras.Sequential over the layers of polus_amd.layers."""

    def __init__(self, layers, compute_dtype="f32", name=None, input_dim=None):
        super().__init__(name)
        self.layers = list(layers)
        self.compute_dtype = COMPUTE_DTYPES[compute_dtype]
        self.arena = ParamArena(self.compute_dtype)
        feat = input_dim
        for i, l in enumerate(self.layers):
            feat = l.build(self.arena, feat, f"{l.name}{i}")
        # a classifier's last Dense feeds a loss: keep its logits in f32
        last_dense = [l for l in self.layers if isinstance(l, Dense)]
        if last_dense and last_dense[-1].activation is None and last_dense[-1].out_dtype is None:
            last_dense[-1].out_dtype = torch.float32
        self.arena.finalize()

    def call(self, x, training=False, **kw):
        x = to_device(x, None, self.arena.device)
        if x.is_floating_point() and x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)
        for l in self.layers:
            x = l.forward(x, training=training)
        return x

    def backward(self, dy, accumulate=False):
        first_param_layer = next(i for i, l in enumerate(self.layers) if l.variables())
        for i in range(len(self.layers) - 1, -1, -1):
            l = self.layers[i]
            if isinstance(l, Dense):
                dy = l.backward(dy, accumulate, need_dx=(i > first_param_layer))
            else:
                dy = l.backward(dy, accumulate)
            self._notify(l.variables())
            if dy is None:
                break
        return dy


class SequentialSavableModel(Sequential, SavableModel):
    pass


class SequentialPolusClassifier(Sequential, PolusClassifier):
    """polus/models.py:152-154 (tutorials/classifier_example.py:44-48)."""
    pass


class BertModel(SavableModel):
    """Embeddings + encoder (+ optional HF pooler, + optional token-level Dense head).  Input protocol of
    the HF model the reference calls: input_ids, attention_mask, token_type_ids (int32 [B,S]).

    `add_pooling_layer=True` adds HF's BertPooler: pooler_output = tanh(W h[:, 0] + b), what the
    reference reads from an unsplit TFBertModel / TFAutoModel (polus/data.py:526-543); without it
    pooler_output is the raw [CLS] state, TFBertSplited's convention (polus/models.py:215-216).
    A SavableModel: `save` writes .cfg (the BertConfig + dtype + head size) and the weights of encoder,
    pooler and head; `polus_amd.models.load_model` rebuilds it through `bert_model`."""

    def __init__(self, cfg, compute_dtype="bf16", num_labels=None, seed=1234, name="bert", with_embeddings=True,
                 layer_indices=None, arena=None, add_pooling_layer=False):
        super().__init__(name)
        self.config = cfg
        self.compute_dtype = COMPUTE_DTYPES[compute_dtype]
        self.savable_config = {"func_name": "bert_model",
                               "model": dict(cfg.to_dict(), compute_dtype="bf16" if self.compute_dtype == torch.bfloat16 else "f32",
                                             num_labels=num_labels, seed=seed, add_pooling_layer=add_pooling_layer)}
        own = arena is None
        self.arena = arena or ParamArena(self.compute_dtype)
        rng = np.random.Generator(np.random.PCG64(seed))
        self.embeddings = BertEmbeddings(cfg, self.arena, rng) if with_embeddings else None
        idx = range(cfg.num_hidden_layers) if layer_indices is None else layer_indices
        self.layer = [BertLayer(cfg, self.arena, i, rng) for i in idx]
        self.pooler = None
        if add_pooling_layer:
            self.pooler = Dense(cfg.hidden_size, activation="tanh", name="pooler")
            self.pooler.build(self.arena, cfg.hidden_size, "pooler")
        self.head = None
        if num_labels:
            self.head = Dense(num_labels, out_dtype=torch.float32, name="head")
            self.head.build(self.arena, cfg.hidden_size, "head")
        if own:
            self.arena.finalize()
        self._scratch = {}
        self.dropout_base_seed = seed
        self.dropout_step = 0     # advanced once per training forward: fresh masks every step
        self.overlap_dw = os.environ.get("POLUS_OVERLAP_DW", "1") != "0"
        self._side = None

    def scratch(self, key, shape, dtype=None):
        dtype = dtype or self.compute_dtype
        t = self._scratch.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype:
            t = self._scratch[key] = torch.empty(tuple(shape), dtype=dtype, device=self.arena.device)
        return t

    def load_numpy_params(self, params, head_w=None, head_b=None):
        """params: dict in oracle/bert.py naming (emb.*, layer{i}.*)."""
        for v in self.arena.vars:
            if v.name in params:
                v.assign(params[v.name])
        if self.head is not None and head_w is not None:
            self.head.w.assign(head_w)
            self.head.b.assign(head_b)

    def site_seed(self, layer, site, step=None):
        """Seed of one dropout site of this step.  In a data-parallel run every rank draws its own masks
        (Horovod/TF ranks have independent RNG streams): the rank is folded into the base seed unless
        `identical_dropout_across_ranks` is set (exact-mask parity tests)."""
        base = self.dropout_base_seed
        if not getattr(self, "identical_dropout_across_ranks", False):
            from . import comm
            if comm.size() > 1:
                base = (base ^ (comm.rank() * 0x9E3779B1)) & 0xFFFFFFFF
        if getattr(self, "graph_seeds", False):     # the step term comes from device memory (polus_amd/graph.py)
            return dropout_seed_static(base, layer, site)
        return dropout_seed(base, self.dropout_step if step is None else step, layer, site)

    def encode(self, hidden, attention_mask, B, S, training=False):
        cfg = self.config
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        for l in self.layer:
            seeds = tuple(self.site_seed(l.index, k) for k in range(3))
            hidden = l.forward(hidden, attention_mask, B, S, p_hid, p_att, seeds, for_backward=training)
        return hidden

    def call(self, input_ids=None, attention_mask=None, token_type_ids=None, training=False, hidden_states=None, **kw):
        dev = self.arena.device
        if isinstance(input_ids, dict):
            d = input_ids
            input_ids, attention_mask = d.get("input_ids"), d.get("attention_mask", attention_mask)
            token_type_ids = d.get("token_type_ids", token_type_ids)
        if attention_mask is not None:
            attention_mask = to_device(attention_mask, torch.int32, dev)
        if hidden_states is None:
            input_ids = to_device(input_ids, torch.int32, dev)
            B, S = input_ids.shape
            if token_type_ids is not None:
                token_type_ids = to_device(token_type_ids, torch.int32, dev)
            hidden = self.embeddings.forward(input_ids, token_type_ids, self.compute_dtype,
                                             self.config.hidden_dropout_prob if training else 0.0, self.site_seed(-1, 0))
        else:
            hs = to_device(hidden_states, self.compute_dtype, dev)
            B, S = hs.shape[0], hs.shape[1]
            hidden = hs.reshape(B * S, -1)
        self._shape = (B, S)
        self.tokens_per_step = B * S          # training.py: whether the optimizer update rides inside backward
        hidden = self.encode(hidden, attention_mask, B, S, training)
        if training:
            self.dropout_step += 1
        H = self.config.hidden_size
        if self.head is not None:
            return self.head.forward(hidden).view(B, S, -1)
        h3 = hidden.view(B, S, H)
        if self.pooler is not None:
            return BaseModelOutputWithPooling(last_hidden_state=h3, pooler_output=self.pooler.forward(h3[:, 0, :], training))
        return BaseModelOutputWithPooling(last_hidden_state=h3, pooler_output=h3[:, 0, :])

    def backward(self, dy=None, accumulate=False, dpooled=None):
        """dy: gradient wrt the logits [B,S,C] (head) or wrt last_hidden_state [B,S,H] (None = zero);
        dpooled: gradient wrt pooler_output [B,H] (models with the HF pooler, or the raw [CLS] slice)."""
        B, S = self._shape
        H = self.config.hidden_size
        if self.head is not None:
            dy = self.head.backward(dy.reshape(B * S, -1), accumulate)
            self._notify(self.head.variables())
        else:
            if dy is None:
                dy = torch.zeros((B * S, H), dtype=self.compute_dtype, device=self.arena.device)
            else:
                dy = to_device(dy, self.compute_dtype, self.arena.device).reshape(B * S, H)
            if dpooled is not None:
                dy = dy.clone()                      # the [CLS] rows are rewritten below: leave the caller's tensor alone
                d3 = dy.view(B, S, H)
                dp = to_device(dpooled, self.compute_dtype, self.arena.device).reshape(B, H)
                if self.pooler is not None:
                    # d h[:,0] = dpooled-through-the-pooler + dy[:,0]: the residual rides on the dX GEMM's epilogue
                    dcls = self.pooler.backward(dp, accumulate, dx_resid=d3[:, 0, :])
                    self._notify(self.pooler.variables())
                else:
                    dcls = self.scratch("dcls", (B, H))
                    ops.gemm(dp, _identity(H, self.compute_dtype, self.arena.device), dcls, resid=d3[:, 0, :])
                d3[:, 0, :].copy_(dcls)
        if self.overlap_dw and self._side is None:
            self._side = torch.cuda.Stream(device=self.arena.device)
        side = self._side if self.overlap_dw else None
        pending = []          # layers whose grouped dW launch is still running on the side stream
        main = torch.cuda.current_stream()

        def retire(lay):
            if lay.dw_done is not None:
                main.wait_event(lay.dw_done)
            self._notify(lay.variables())
        for l in reversed(self.layer):
            if len(pending) == 2:      # this layer reuses the dY buffers of the layer two steps back
                retire(pending.pop(0))
            dy = l.backward(dy, self.scratch, accumulate, side)
            pending.append(l)
        for lay in pending:
            retire(lay)
        if self.embeddings is not None:
            self.embeddings.backward(dy, accumulate, self.deterministic)
            self._notify(self.embeddings.variables())
            return None
        return dy.view(B, S, -1)

    def inference(self, x):
        logits = self(**x, training=False) if isinstance(x, dict) else self(x, training=False)
        l2 = logits.reshape(-1, logits.shape[-1]).contiguous()
        out = torch.empty(l2.shape[0], dtype=torch.int32, device=l2.device)
        ops.argmax(l2, out)
        return out.view(logits.shape[:-1])


_EYE = {}


def _identity(n, dtype, dev):
    k = (n, dtype, str(dev))
    if k not in _EYE:
        _EYE[k] = torch.eye(n, dtype=dtype, device=dev)
    return _EYE[k]


@from_config
def bert_model(**kw):
    """Builder behind BertModel.save / load_model: BertConfig fields + compute_dtype, num_labels, seed,
    add_pooling_layer."""
    cfg_keys = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                "max_position_embeddings", "type_vocab_size", "layer_norm_eps", "hidden_dropout_prob",
                "attention_probs_dropout_prob", "_name_or_path")
    cfg = BertConfig(**{k: kw[k] for k in cfg_keys if k in kw})
    return BertModel(cfg, compute_dtype=kw.get("compute_dtype", "bf16"), num_labels=kw.get("num_labels"),
                     seed=kw.get("seed", 1234), add_pooling_layer=kw.get("add_pooling_layer", False))


class TFBertSplited(PolusModel):
    """polus/models.py:164-216: runs a slice of encoder layers over given hidden states with
    the (1-m)*-10000 mask (applied inside the attention kernel) and returns
    pooler_output = hidden[:, 0, :] (no dense, no tanh).  Shares the layer objects — and so
    the weights — of the model it was split from."""

    def __init__(self, bert_layers, arena, config, run_in_training_mode=True, name="bert_splited"):
        super().__init__(name)
        self.layer = list(bert_layers)
        self.arena, self.config = arena, config
        self.run_in_training_mode = run_in_training_mode
        self.compute_dtype = arena.compute_dtype
        self._scratch = {}
        self.dropout_base_seed, self.dropout_step = 4321, 0

    scratch = BertModel.scratch
    site_seed = BertModel.site_seed

    @property
    def trainable_weights(self):
        return [v for l in self.layer for v in l.variables()]

    def call(self, hidden_states, attention_mask, training=False, **kw):
        dev = self.arena.device
        hs = to_device(hidden_states, self.compute_dtype, dev)
        B, S, H = hs.shape
        mask = to_device(attention_mask, torch.int32, dev)
        hidden = hs.reshape(B * S, H)
        train = bool(self.run_in_training_mode and training)      # polus/models.py:213
        p_hid = self.config.hidden_dropout_prob if train else 0.0
        p_att = self.config.attention_probs_dropout_prob if train else 0.0
        for l in self.layer:
            hidden = l.forward(hidden, mask, B, S, p_hid, p_att, tuple(self.site_seed(l.index, k) for k in range(3)), for_backward=bool(training))
        if train:
            self.dropout_step += 1
        self._shape = (B, S)
        self.tokens_per_step = B * S          # training.py: whether the optimizer update rides inside backward
        h3 = hidden.view(B, S, H)
        return BaseModelOutputWithPooling(last_hidden_state=h3, pooler_output=h3[:, 0, :])

    def backward(self, dy, accumulate=False):
        B, S = self._shape
        dy = to_device(dy, self.compute_dtype, self.arena.device).reshape(B * S, -1)
        for l in reversed(self.layer):
            dy = l.backward(dy, self.scratch, accumulate)
            self._notify(l.variables())
        return dy.view(B, S, -1)


def split_bert_model(bert_model, index_layer, init_models=False, return_pre_bert_model=True,
                     return_post_bert_model=True):
    """polus/models.py:242-295: cut `bert_model` at `index_layer` (negative allowed,
    != 0, |index| < L); the pre model keeps the embeddings and the first layers, the post model
    (TFBertSplited) runs the remaining ones on the same weights."""
    assert return_pre_bert_model or return_post_bert_model
    L = bert_model.config.num_hidden_layers
    assert L > index_layer > -L and index_layer != 0
    post_model = None
    if return_post_bert_model:
        post_model = TFBertSplited(bert_model.layer[index_layer:], bert_model.arena, bert_model.config)
    if return_pre_bert_model:
        del bert_model.layer[index_layer:]
        bert_model.config.num_hidden_layers = len(bert_model.layer)
    if init_models and return_pre_bert_model:
        B, S = 1, min(50, bert_model.config.max_position_embeddings)
        ids = np.ones((B, S), np.int32)
        out = bert_model(input_ids=ids, attention_mask=np.ones((B, S), np.int32), token_type_ids=np.zeros((B, S), np.int32))
        if post_model is not None:
            post_model(hidden_states=out.last_hidden_state, attention_mask=np.ones((B, S), np.int32))
    if return_pre_bert_model and return_post_bert_model:
        return bert_model, post_model
    return bert_model if return_pre_bert_model else post_model


def split_bert_model_from_checkpoint(bert_model_checkpoint, index_layer, **kw):
    """polus/models.py:219-240 fetches the checkpoint by name from the HF hub; there is no
    network here, so only a local directory holding config.json + *.safetensors is accepted."""
    from .checkpoint import load_bert_from_local
    return split_bert_model(load_bert_from_local(bert_model_checkpoint), index_layer, **kw)
