"""Weights in and out of the engine.

* HF PyTorch BERT checkpoints from a LOCAL directory (config.json + model.safetensors): the
  reference fetches them by name with `from_pt=True` (polus/models.py:225-229); there is no network
  here, so only local paths are accepted.  `safetensors` executes nothing from the file.
* SavableModel files written by `SavableModel.save` (polus/models.py:112-133 analogue):
  <name>.cfg (JSON) + <name>.npz with weight0..N in get_weights() order.
"""
import json
import os

import numpy as np

HF_LAYER_MAP = [
    ("attention.output.dense.weight", "out.w"), ("attention.output.dense.bias", "out.b"),
    ("attention.output.LayerNorm.weight", "ln1.g"), ("attention.output.LayerNorm.bias", "ln1.b"),
    ("intermediate.dense.weight", "ffn1.w"), ("intermediate.dense.bias", "ffn1.b"),
    ("output.dense.weight", "ffn2.w"), ("output.dense.bias", "ffn2.b"),
    ("output.LayerNorm.weight", "ln2.g"), ("output.LayerNorm.bias", "ln2.b"),
]


def hf_state_to_params(state, num_layers):
    """HF BertModel state dict (name -> ndarray; an optional 'bert.' prefix is stripped, old
    'gamma'/'beta' LayerNorm names accepted) -> this repo's names with Q/K/V fused into qkv.{w,b}."""
    st = {}
    for k, v in state.items():
        k = k[5:] if k.startswith("bert.") else k
        k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
        st[k] = np.asarray(v)
    p = {"emb.word": st["embeddings.word_embeddings.weight"], "emb.pos": st["embeddings.position_embeddings.weight"],
         "emb.type": st["embeddings.token_type_embeddings.weight"], "emb.ln.g": st["embeddings.LayerNorm.weight"],
         "emb.ln.b": st["embeddings.LayerNorm.bias"]}
    for i in range(num_layers):
        q, o = f"encoder.layer.{i}.", f"layer{i}."
        p[o + "qkv.w"] = np.concatenate([st[q + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
        p[o + "qkv.b"] = np.concatenate([st[q + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
        for hf, ours in HF_LAYER_MAP:
            p[o + ours] = st[q + hf]
    if "pooler.dense.weight" in st:
        p["pooler.w"], p["pooler.b"] = st["pooler.dense.weight"], st["pooler.dense.bias"]
    if "classifier.weight" in state:
        p["head.w"], p["head.b"] = np.asarray(state["classifier.weight"]), np.asarray(state["classifier.bias"])
    return p


def read_local_hf_checkpoint(path):
    """(config dict, params dict) from a local HF BERT directory."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    from safetensors.numpy import load_file
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"{path}: no *.safetensors file (pickled pytorch_model.bin files are not loaded)")
    state = {}
    for f in files:
        state.update(load_file(os.path.join(path, f)))
    return cfg, hf_state_to_params(state, cfg["num_hidden_layers"])


def load_bert_from_local(path, compute_dtype="bf16", num_labels=None, add_pooling_layer=False):
    from .models import BertConfig, BertModel
    if not os.path.isdir(path):
        raise FileNotFoundError(f"'{path}' is not a local directory: checkpoints cannot be fetched by name (no network)")
    c, params = read_local_hf_checkpoint(path)
    cfg = BertConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                     num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                     max_position_embeddings=c["max_position_embeddings"], type_vocab_size=c.get("type_vocab_size", 2),
                     layer_norm_eps=c.get("layer_norm_eps", 1e-12),
                     hidden_dropout_prob=c.get("hidden_dropout_prob", 0.1),
                     attention_probs_dropout_prob=c.get("attention_probs_dropout_prob", 0.1), _name_or_path=path)
    model = BertModel(cfg, compute_dtype=compute_dtype, num_labels=num_labels, add_pooling_layer=add_pooling_layer)
    model.load_numpy_params(params)
    return model


def load_weights(model, path):
    """Inverse of SavableModel.save: <path>.npz -> model.set_weights (weight{i} order = get_weights())."""
    z = np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False)
    model.set_weights([z[f"weight{i}"] for i in range(len(z.files))])
    return model


# ------------------------------------------------------------------------------------ resume state
def save_training_state(trainer, path):
    """Everything a run needs to continue bit-for-bit after a restart, which the reference's
    SaveModelCallback does not keep (polus/callbacks.py:264-313 saves weights only): the f32 master
    parameters of every arena the trainer updates, the Adam moments, the optimizer's iteration count
    (the learning-rate schedule's argument), the trainer's step / micro-step counters and the models'
    dropout counters.  One `<path>.state.npz`; written by rank 0's caller.  Data-parallel runs on the
    reduce-scatter scheme keep each rank's Adam moments current only on the slices it owns: every rank calls
    `trainer.sync_optimizer_state()` (a collective) first, then rank 0 saves -- saving unsynchronised moments
    is refused rather than written."""
    if getattr(trainer, "use_horovod", False) and not getattr(trainer, "_opt_state_synced", True):
        raise RuntimeError("save_training_state: this rank's optimizer moments are current only on its own arena slices "
                           "(data-parallel reduce-scatter scheme); call trainer.sync_optimizer_state() on every rank first")
    arenas = trainer._arenas()
    opt = trainer.optimizer
    out = {"n_arenas": np.int64(len(arenas)), "iterations": np.int64(getattr(opt, "iterations", 0)),
           "step_counter": np.int64(trainer.step_counter), "micro": np.int64(getattr(trainer, "step_counter_micro", 0)),
           "dropout_step": np.int64(getattr(trainer.model, "dropout_step", 0))}
    for i, a in enumerate(arenas):
        n = getattr(a, "extent", a.params.numel())       # the padding behind the last variable is not state
        out[f"a{i}.params"] = a.params[:n].detach().cpu().numpy()
        if hasattr(opt, "_slots"):
            m, v = opt._slots(a)
            out[f"a{i}.m"], out[f"a{i}.v"] = m[:n].detach().cpu().numpy(), v[:n].detach().cpu().numpy()
        if getattr(trainer, "grad_accum_steps", 1) > 1:
            out[f"a{i}.grads"] = a.grads[:n].detach().cpu().numpy()      # a partially accumulated step
    p = path if path.endswith(".state.npz") else path + ".state.npz"
    np.savez(p, **out)
    return p


def load_training_state(trainer, path):
    """Inverse of save_training_state on a trainer built the same way (same model geometry and dtype)."""
    import torch
    p = path if path.endswith(".state.npz") else path + ".state.npz"
    z = np.load(p, allow_pickle=False)
    arenas = trainer._arenas()
    if int(z["n_arenas"]) != len(arenas):
        raise ValueError(f"{p}: saved from {int(z['n_arenas'])} parameter arenas, this trainer has {len(arenas)}")
    opt = trainer.optimizer
    for i, a in enumerate(arenas):
        w = z[f"a{i}.params"]
        n = getattr(a, "extent", a.params.numel())
        if w.shape[0] < n or w.shape[0] > a.params.numel() and np.any(w[a.params.numel():]):
            raise ValueError(f"{p}: arena {i} holds {w.shape[0]} parameters, the model has {n}")
        n = min(w.shape[0], a.params.numel())            # files of earlier versions carry their (zero) padding

        def put(dst, src):
            dst[:n].copy_(torch.from_numpy(src[:n]))
            dst[n:].zero_()
        put(a.params, w)
        a.refresh_shadow()
        if f"a{i}.m" in z.files and hasattr(opt, "_slots"):
            m, v = opt._slots(a)
            put(m, z[f"a{i}.m"])
            put(v, z[f"a{i}.v"])
        if f"a{i}.grads" in z.files:
            put(a.grads, z[f"a{i}.grads"])
    if hasattr(opt, "iterations"):
        opt.iterations = int(z["iterations"])
    trainer.step_counter = int(z["step_counter"])
    trainer.step_counter_micro = int(z["micro"])
    if hasattr(trainer.model, "dropout_step"):
        trainer.model.dropout_step = int(z["dropout_step"])
    return trainer
