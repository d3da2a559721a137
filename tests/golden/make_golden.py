"""Generates tests/golden/*.npz from the PyTorch-CPU `transformers` BERT.

Why this source: the reference (bioinformatics-ua/polus) runs HuggingFace TF-BERT loaded
with from_pt=True (polus/models.py:225-229); TensorFlow is not installed here and the
reference's tests hold no numeric vectors for this path, so the vectors come from the
PyTorch twin of that model (same architecture, same weight format), built from a local
BertConfig with seeded weights — nothing is downloaded.  The additive mask follows
polus/models.py:181-195 ((1-m) * -10000), not HF's finfo.min.

Run once in the build container:   python tests/golden/make_golden.py
Only data (inputs + expected outputs) is written; this script is the recipe.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bert as ob  # noqa: E402
from oracle import optim as oo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hf_model(cfg, params, dtype=torch.float64):
    from transformers import BertConfig, BertModel
    hc = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                    num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads,
                    intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings,
                    type_vocab_size=cfg.type_vocab_size,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                    layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu",
                    pad_token_id=None)  # TF gather has no padding_idx: row 0 gets its gradient
    hc._attn_implementation = "eager"
    m = BertModel(hc, add_pooling_layer=False).to(dtype)
    H = cfg.hidden_size
    sd = {}
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    sd["embeddings.word_embeddings.weight"] = t(params["emb.word"])
    sd["embeddings.position_embeddings.weight"] = t(params["emb.pos"])
    sd["embeddings.token_type_embeddings.weight"] = t(params["emb.type"])
    sd["embeddings.LayerNorm.weight"] = t(params["emb.ln.g"])
    sd["embeddings.LayerNorm.bias"] = t(params["emb.ln.b"])
    for i in range(cfg.num_hidden_layers):
        p, q = f"layer{i}.", f"encoder.layer.{i}."
        w, b = params[p + "qkv.w"], params[p + "qkv.b"]
        for j, nm in enumerate(["query", "key", "value"]):
            sd[q + f"attention.self.{nm}.weight"] = t(w[j * H:(j + 1) * H])
            sd[q + f"attention.self.{nm}.bias"] = t(b[j * H:(j + 1) * H])
        sd[q + "attention.output.dense.weight"] = t(params[p + "out.w"])
        sd[q + "attention.output.dense.bias"] = t(params[p + "out.b"])
        sd[q + "attention.output.LayerNorm.weight"] = t(params[p + "ln1.g"])
        sd[q + "attention.output.LayerNorm.bias"] = t(params[p + "ln1.b"])
        sd[q + "intermediate.dense.weight"] = t(params[p + "ffn1.w"])
        sd[q + "intermediate.dense.bias"] = t(params[p + "ffn1.b"])
        sd[q + "output.dense.weight"] = t(params[p + "ffn2.w"])
        sd[q + "output.dense.bias"] = t(params[p + "ffn2.b"])
        sd[q + "output.LayerNorm.weight"] = t(params[p + "ln2.g"])
        sd[q + "output.LayerNorm.bias"] = t(params[p + "ln2.b"])
    missing = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    return m


def hf_grads_to_ours(m, cfg):
    H = cfg.hidden_size
    g = {}
    sd = {n: p.grad.numpy() for n, p in m.named_parameters()}
    g["emb.word"] = sd["embeddings.word_embeddings.weight"]
    g["emb.pos"] = sd["embeddings.position_embeddings.weight"]
    g["emb.type"] = sd["embeddings.token_type_embeddings.weight"]
    g["emb.ln.g"] = sd["embeddings.LayerNorm.weight"]
    g["emb.ln.b"] = sd["embeddings.LayerNorm.bias"]
    for i in range(cfg.num_hidden_layers):
        p, q = f"layer{i}.", f"encoder.layer.{i}."
        g[p + "qkv.w"] = np.concatenate([sd[q + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
        g[p + "qkv.b"] = np.concatenate([sd[q + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
        g[p + "out.w"] = sd[q + "attention.output.dense.weight"]
        g[p + "out.b"] = sd[q + "attention.output.dense.bias"]
        g[p + "ln1.g"] = sd[q + "attention.output.LayerNorm.weight"]
        g[p + "ln1.b"] = sd[q + "attention.output.LayerNorm.bias"]
        g[p + "ffn1.w"] = sd[q + "intermediate.dense.weight"]
        g[p + "ffn1.b"] = sd[q + "intermediate.dense.bias"]
        g[p + "ffn2.w"] = sd[q + "output.dense.weight"]
        g[p + "ffn2.b"] = sd[q + "output.dense.bias"]
        g[p + "ln2.g"] = sd[q + "output.LayerNorm.weight"]
        g[p + "ln2.b"] = sd[q + "output.LayerNorm.bias"]
    return g


def synth_batch(cfg, B, S, C, seed):
    """SURVEY.md §8(d) synthetic inputs: ids U{lo..V-1}, CLS first, SEP last-valid,
    ragged lengths U{S/2..S}, labels U{0..C-1}, padded positions label 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lo = min(1000, cfg.vocab_size // 2)
    ids = rng.integers(lo, cfg.vocab_size, size=(B, S)).astype(np.int32)
    lens = rng.integers(S // 2, S + 1, size=(B,))
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    ids[:, 0] = min(101, cfg.vocab_size - 1)
    for b in range(B):
        ids[b, lens[b] - 1] = min(102, cfg.vocab_size - 1)
    ids = ids * mask
    labels = (rng.integers(0, C, size=(B, S)) * mask).astype(np.int32)
    tt = np.zeros_like(ids)
    return ids, mask, tt, labels


def hf_step(m, head_w, head_b, ids, mask, tt, labels):
    for p in m.parameters():
        p.grad = None
    hw = torch.tensor(head_w, dtype=torch.float64, requires_grad=True)
    hb = torch.tensor(head_b, dtype=torch.float64, requires_grad=True)
    emb = m.embeddings(input_ids=torch.tensor(ids, dtype=torch.long),
                       token_type_ids=torch.tensor(tt, dtype=torch.long))
    add = (1.0 - torch.tensor(mask, dtype=torch.float64))[:, None, None, :] * -10000.0
    last = m.encoder(emb, attention_mask=add).last_hidden_state
    logits = last @ hw.T + hb
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]),
                                             torch.tensor(labels, dtype=torch.long).reshape(-1))
    loss.backward()
    return loss.item(), logits.detach().numpy(), last.detach().numpy(), hw.grad.numpy(), hb.grad.numpy()


def make_case(name, cfg, B, S, C, seed, steps=0):
    params, head_w, head_b = ob.golden_setup(cfg, C)
    ids, mask, tt, labels = synth_batch(cfg, B, S, C, seed)
    m = hf_model(cfg, params)
    loss, logits, last, ghw, ghb = hf_step(m, head_w, head_b, ids, mask, tt, labels)
    g = hf_grads_to_ours(m, cfg)

    # cross-check the NumPy restatement against the HF twin in float64
    oloss, ologits, cache = ob.token_classifier_fwd(params, cfg, head_w, head_b, ids, mask, labels, tt)
    og = ob.token_classifier_bwd(params, cfg, head_w, cache)
    assert abs(oloss - loss) < 1e-10, (oloss, loss)
    assert np.abs(ologits - logits).max() < 1e-9
    for k in g:
        err = np.abs(og[k] - g[k]).max()
        assert err < 1e-9, (k, err)
    assert np.abs(og["head.w"] - ghw).max() < 1e-10

    out = dict(ids=ids, mask=mask, token_type=tt, labels=labels,
               head_w=head_w.astype(np.float32), head_b=head_b.astype(np.float32),
               loss=np.float64(loss), logits=logits.astype(np.float32),
               last_hidden=last.astype(np.float32),
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers,
                             cfg.num_attention_heads, cfg.intermediate_size,
                             cfg.max_position_embeddings, cfg.type_vocab_size], np.int64),
               param_perturb_seed=np.int64(77))
    # gradients: per-tensor L2 norm + first 64 entries (fixtures stay small)
    names = sorted(g)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([np.sqrt((g[k] ** 2).sum()) for k in names])
    out["grad_heads"] = np.stack([np.resize(g[k].reshape(-1)[:64], 64) for k in names]).astype(np.float32)
    out["grad_head_w"] = ghw.astype(np.float32)
    out["param_checksum"] = np.array([float(np.abs(params[k]).sum()) for k in sorted(params)])

    if steps:
        # loss trajectory: HF gradients + the restated Keras-Adam/AdamWeightDecay update
        opt = oo.Adam(lr=lambda t: oo.warmup_linear_lr(t, steps, 1e-3), weight_decay=0.01,
                      no_decay=[k for k in list(params) + ["head.w", "head.b"] if oo.is_no_decay(k)])
        allp = dict(params); allp["head.w"] = head_w; allp["head.b"] = head_b
        traj = []
        for s in range(steps):
            bi, bm, bt, bl = synth_batch(cfg, B, S, C, seed + s)
            m = hf_model(cfg, allp)
            l, _, _, ghw, ghb = hf_step(m, allp["head.w"], allp["head.b"], bi, bm, bt, bl)
            gg = hf_grads_to_ours(m, cfg); gg["head.w"] = ghw; gg["head.b"] = ghb
            traj.append(l)
            opt.step(allp, gg)
        out["traj_loss"] = np.array(traj)
        out["traj_steps"] = np.int64(steps)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss, "ok")


if __name__ == "__main__":
    torch.manual_seed(0)
    small = ob.BertConfig(vocab_size=96, hidden_size=128, num_hidden_layers=2,
                          num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=64, type_vocab_size=2)
    make_case("bert_small_b2_s16", small, B=2, S=16, C=4, seed=42, steps=5)
    make_case("bert_small_b3_s48", small, B=3, S=48, C=4, seed=43)
    base1 = ob.BertConfig(vocab_size=512, hidden_size=768, num_hidden_layers=1,
                          num_attention_heads=12, intermediate_size=3072,
                          max_position_embeddings=128, type_vocab_size=2)
    make_case("bert_base1_b2_s64", base1, B=2, S=64, C=4, seed=44)
