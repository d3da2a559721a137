"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product (polus_amd/).

torch-CPU restatement of the same training step the NumPy oracle (oracle/bert.py, oracle/optim.py)
restates: BERT encoder + token-classification head forward (the op graph of
the HF TF-BERT layers that polus/models.py:201-216 drives), mean sparse softmax cross-entropy
(Keras SparseCategoricalCrossentropy(from_logits=True), tutorials/classifier_example.py:55), the
gradients (torch autograd here, explicit in oracle/bert.py) and the Keras-Adam / HF-AdamWeightDecay
update of oracle/optim.py.  Same parameter names and layouts as oracle/bert.py (weights [out, in],
Q/K/V fused into qkv.w [3H, H]).

Two uses: (1) bench.py's `cpu_baseline` leg times it on the GPU box's host cores (SURVEY.md §8(d):
torch-CPU eager fp32, all cores) -- the NumPy oracle spends its time in single-threaded elementwise
passes and is ~10x slower, which says nothing about the host; (2) tests/test_oracle_golden.py checks
it against the NumPy oracle, so the thing timed is the thing pinned.
"""
import math

import torch


def to_torch(params, dtype=torch.float32):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)            # biased variance, eps inside the sqrt
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))  # exact erf form (HF hidden_act="gelu")


def token_classifier_loss(p, cfg, input_ids, attention_mask, labels, token_type_ids=None):
    """loss (mean over every position, padded ones included -- tf.reduce_mean over all leading dims), logits."""
    ids = torch.as_tensor(input_ids, dtype=torch.long)
    B, S = ids.shape
    H, A = cfg.hidden_size, cfg.num_attention_heads
    d = H // A
    tt = torch.zeros_like(ids) if token_type_ids is None else torch.as_tensor(token_type_ids, dtype=torch.long)
    x = p["emb.word"][ids] + p["emb.pos"][:S][None] + p["emb.type"][tt]
    x = _ln(x, p["emb.ln.g"], p["emb.ln.b"], cfg.layer_norm_eps).reshape(B * S, H)
    m = torch.as_tensor(attention_mask).to(x.dtype)
    add_mask = ((1.0 - m) * -10000.0)[:, None, None, :]     # polus/models.py:190-193
    for i in range(cfg.num_hidden_layers):
        q = f"layer{i}."
        qkv = x @ p[q + "qkv.w"].T + p[q + "qkv.b"]
        qh, kh, vh = (t.reshape(B, S, A, d).permute(0, 2, 1, 3) for t in qkv.split(H, dim=1))
        sc = qh @ kh.transpose(-1, -2) / math.sqrt(d) + add_mask
        ctx = (torch.softmax(sc, -1) @ vh).permute(0, 2, 1, 3).reshape(B * S, H)
        a1 = _ln(ctx @ p[q + "out.w"].T + p[q + "out.b"] + x, p[q + "ln1.g"], p[q + "ln1.b"], cfg.layer_norm_eps)
        f = _gelu(a1 @ p[q + "ffn1.w"].T + p[q + "ffn1.b"])
        x = _ln(f @ p[q + "ffn2.w"].T + p[q + "ffn2.b"] + a1, p[q + "ln2.g"], p[q + "ln2.b"], cfg.layer_norm_eps)
    logits = x @ p["head.w"].T + p["head.b"]
    lab = torch.as_tensor(labels, dtype=torch.long).reshape(-1)
    loss = torch.nn.functional.cross_entropy(logits, lab, reduction="mean")
    return loss, logits.reshape(B, S, -1)


class Adam:
    """oracle/optim.py Adam in torch: Keras form (eps outside the bias-corrected sqrt, default 1e-7) +
    HF AdamWeightDecay's decoupled decay on everything but LayerNorm / bias."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, weight_decay=0.0, no_decay=()):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.no_decay = set(no_decay)
        self.t = 0
        self.m, self.v = {}, {}

    @torch.no_grad()
    def step(self, p):
        lr = self.lr(self.t) if callable(self.lr) else self.lr
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, w in p.items():
            g = w.grad
            if g is None:
                continue
            if k not in self.m:
                self.m[k], self.v[k] = torch.zeros_like(w), torch.zeros_like(w)
            if self.wd and k not in self.no_decay:
                w.mul_(1.0 - lr * self.wd)
            self.m[k].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            w.addcdiv_(self.m[k], self.v[k].sqrt().add_(self.eps), value=-lr_t)
            w.grad = None


def train_step(p, cfg, opt, input_ids, attention_mask, labels, token_type_ids=None):
    loss, _ = token_classifier_loss(p, cfg, input_ids, attention_mask, labels, token_type_ids)
    loss.backward()
    opt.step(p)
    return float(loss.detach())
