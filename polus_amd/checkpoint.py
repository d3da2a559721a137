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


def load_bert_from_local(path, compute_dtype="bf16", num_labels=None):
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
    model = BertModel(cfg, compute_dtype=compute_dtype, num_labels=num_labels)
    model.load_numpy_params(params)
    return model


def load_weights(model, path):
    """Inverse of SavableModel.save: <path>.npz -> model.set_weights (weight{i} order = get_weights())."""
    z = np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False)
    model.set_weights([z[f"weight{i}"] for i in range(len(z.files))])
    return model
