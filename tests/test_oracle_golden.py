"""CPU: the NumPy oracle against the committed golden vectors (HF PyTorch twin, float64).
This is what pins the oracle; the GPU tests then pin the HIP engine to the oracle."""
import os

import numpy as np
import pytest

from oracle import bert as ob
from oracle import losses as ol
from oracle import optim as oo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["bert_small_b2_s16", "bert_small_b3_s48", "bert_base1_b2_s64"]


def load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    v, h, l, a, i, p, t = [int(x) for x in g["cfg"]]
    cfg = ob.BertConfig(v, h, l, a, i, p, t)
    params, hw, hb = ob.golden_setup(cfg, g["logits"].shape[-1])
    return g, cfg, params, hw, hb


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_golden_float64(case):
    g, cfg, params, hw, hb = load(case)
    chk = np.array([float(np.abs(params[k]).sum()) for k in sorted(params)])
    assert np.allclose(chk, g["param_checksum"], rtol=1e-12)
    loss, logits, cache = ob.token_classifier_fwd(params, cfg, hw, hb, g["ids"], g["mask"], g["labels"], g["token_type"])
    assert abs(loss - float(g["loss"])) < 1e-10
    assert np.abs(logits - g["logits"]).max() < 2e-6  # golden logits are stored as f32
    grads = ob.token_classifier_bwd(params, cfg, hw, cache)
    for k, norm, head in zip([str(n) for n in g["grad_names"]], g["grad_norms"], g["grad_heads"]):
        a = grads[k].reshape(-1)
        assert abs(np.sqrt((a ** 2).sum()) - norm) < 1e-9 * max(1, norm), k
        n = min(64, a.size)
        assert np.abs(a[:n] - head[:n]).max() < 1e-6, k


def test_oracle_float32_close_to_golden():
    g, cfg, params, hw, hb = load("bert_small_b3_s48")
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    loss, logits, _ = ob.token_classifier_fwd(p32, cfg, hw.astype(np.float32), hb.astype(np.float32),
                                              g["ids"], g["mask"], g["labels"], g["token_type"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert np.abs(logits - g["logits"]).max() < 1e-4


def test_trajectory_replay():
    """The 5-step AdamW trajectory in the golden file, replayed with oracle gradients."""
    from tests.golden.make_golden import synth_batch
    g, cfg, params, hw, hb = load("bert_small_b2_s16")
    steps = int(g["traj_steps"])
    allp = dict(params); allp["head.w"] = hw; allp["head.b"] = hb
    opt = oo.Adam(lr=lambda t: oo.warmup_linear_lr(t, steps, 1e-3), weight_decay=0.01,
                  no_decay=[k for k in allp if oo.is_no_decay(k)])
    for s in range(steps):
        ids, mask, tt, labels = synth_batch(cfg, 2, 16, 4, 42 + s)
        loss, _, cache = ob.token_classifier_fwd(allp, cfg, allp["head.w"], allp["head.b"], ids, mask, labels, tt)
        assert abs(loss - g["traj_loss"][s]) < 1e-9, s
        opt.step(allp, ob.token_classifier_bwd(allp, cfg, allp["head.w"], cache))


def test_mask_convention_and_pooler_slice():
    """polus/models.py:175-195 and :215-216."""
    m = np.array([[1, 1, 0], [1, 0, 0]], np.int32)
    add = ob.additive_mask(m)
    assert add.shape == (2, 1, 1, 3) and add.dtype == np.float32
    assert np.array_equal(add.reshape(2, 3), np.array([[0, 0, -10000.0], [0, -10000.0, -10000.0]], np.float32))
    cfg = ob.BertConfig(50, 64, 1, 1, 128, 16, 2)
    p = ob.init_params(cfg)
    h = np.random.default_rng(0).standard_normal((2, 3, 64)).astype(np.float32)
    last, pooled, _ = ob.encoder_fwd(p, cfg, h, m)
    assert np.array_equal(pooled, last[:, 0, :])


def test_numeric_gradients_of_primitives():
    r = np.random.default_rng(1)
    x = r.standard_normal((3, 8))
    g_, b_ = 1 + 0.1 * r.standard_normal(8), 0.1 * r.standard_normal(8)
    dy = r.standard_normal((3, 8))
    y, mean, rstd = ob.layer_norm_fwd(x, g_, b_, 1e-12)
    dx, dg, db = ob.layer_norm_bwd(dy, x, g_, mean, rstd)
    eps = 1e-6
    num = np.zeros_like(x)
    for i in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy(); xp[i] += eps; xm[i] -= eps
        num[i] = ((ob.layer_norm_fwd(xp, g_, b_, 1e-12)[0] - ob.layer_norm_fwd(xm, g_, b_, 1e-12)[0]) * dy).sum() / (2 * eps)
    assert np.abs(num - dx).max() < 1e-6
    u = r.standard_normal(100)
    assert np.abs((ob.gelu(u + eps) - ob.gelu(u - eps)) / (2 * eps) - ob.gelu_grad(u)).max() < 1e-6
    assert np.abs((ob.swish(u + eps) - ob.swish(u - eps)) / (2 * eps) - ob.swish_grad(u)).max() < 1e-6


def test_crf_gradients_numeric_and_viterbi_bruteforce():
    r = np.random.default_rng(3)
    B, S, C = 2, 4, 3
    pot = r.standard_normal((B, S, C))
    tags = r.integers(0, C, size=(B, S))
    lens = np.array([4, 3])
    T = r.standard_normal((C, C)) * 0.5
    y = np.eye(C)[tags]
    loss, dx, dT = ol.crf_nll_fwd(y, pot, lens, T)
    eps = 1e-6
    for i in [(0, 0, 0), (0, 3, 2), (1, 2, 1), (1, 3, 0)]:
        pp, pm = pot.copy(), pot.copy(); pp[i] += eps; pm[i] -= eps
        num = (ol.crf_nll_fwd(y, pp, lens, T)[0] - ol.crf_nll_fwd(y, pm, lens, T)[0]) / (2 * eps)
        assert abs(num - dx[i]) < 1e-6, i
    for i in [(0, 0), (1, 2), (2, 1)]:
        Tp, Tm = T.copy(), T.copy(); Tp[i] += eps; Tm[i] -= eps
        num = (ol.crf_nll_fwd(y, pot, lens, Tp)[0] - ol.crf_nll_fwd(y, pot, lens, Tm)[0]) / (2 * eps)
        assert abs(num - dT[i]) < 1e-6, i
    # log-likelihood normalises: sum over all paths of exp(ll) == 1
    import itertools
    for b in range(B):
        L = lens[b]
        tot = 0.0
        best, best_s = None, -1e30
        for path in itertools.product(range(C), repeat=int(L)):
            tg = np.zeros((1, S), np.int64); tg[0, :L] = path
            ll = ol.crf_log_likelihood(pot[b:b + 1], tg, lens[b:b + 1], T)[0][0]
            tot += np.exp(ll)
            if ll > best_s:
                best, best_s = path, ll
        assert abs(tot - 1.0) < 1e-9
        assert tuple(ol.crf_viterbi(pot[b:b + 1], lens[b:b + 1], T)[0, :L]) == best
    # impossible-transition mask (polus/layers.py:58-63)
    M = np.ones((C, C)); M[0, 2] = 0
    Tm_ = ol.crf_transitions(T, M)
    assert Tm_[0, 2] == -10000 and Tm_[1, 1] == T[1, 1]


def test_losses_against_definitions():
    r = np.random.default_rng(5)
    logits = r.standard_normal((6, 4))
    labels = r.integers(0, 4, size=6)
    loss, d = ol.sparse_softmax_xent_fwd(logits, labels)
    p = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    assert abs(loss + np.log(p[np.arange(6), labels]).mean()) < 1e-12
    cw = np.array([1.0, 2.0, 0.5, 3.0])
    oh = np.eye(4)[labels]
    wl, _ = ol.weighted_softmax_xent_fwd(cw, oh, logits)
    assert abs(wl - (-np.log(p[np.arange(6), labels]) * cw[labels]).mean()) < 1e-12
    y = (r.uniform(size=(6, 4)) < 0.4).astype(float); y[0] = 0
    sl, sd = ol.weighted_sigmoid_xent_fwd(cw, 0.25, y, logits)
    sig = 1 / (1 + np.exp(-logits))
    per = -(y * np.log(sig) + (1 - y) * np.log(1 - sig)).sum(-1)
    w = (cw * y).sum(-1) + (y.sum(-1) == 0) * 0.25
    assert abs(sl - (per * w).mean()) < 1e-10


def test_adam_and_schedule_definitions():
    # schedule: polus/schedulers.py:5-23 (end lr hard-coded 1e-7)
    N, lr = 100, 1e-3
    assert oo.warmup_linear_lr(0, N, lr) == 0.0
    assert abs(oo.warmup_linear_lr(5, N, lr) - lr * 0.5) < 1e-15
    assert abs(oo.warmup_linear_lr(10, N, lr) - lr) < 1e-15
    assert abs(oo.warmup_linear_lr(100, N, lr) - 1e-7) < 1e-15
    assert abs(oo.warmup_linear_lr(500, N, lr) - 1e-7) < 1e-15
    assert abs(oo.warmup_linear_lr(55, N, lr) - ((lr - 1e-7) * 0.5 + 1e-7)) < 1e-15
    # Keras Adam first step: p -= lr * g / (|g| + eps*...) ~ lr*sign(g)
    p = {"w": np.array([1.0, -2.0]), "ln.g": np.array([1.0])}
    g = {"w": np.array([0.5, -0.25]), "ln.g": np.array([0.1])}
    opt = oo.Adam(lr=0.1, weight_decay=0.01, no_decay=["ln.g"])
    opt.step(p, g)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    w0 = np.array([1.0, -2.0]) * (1 - 0.1 * 0.01)
    m, v = 0.1 * g["w"], 0.001 * g["w"] ** 2
    assert np.allclose(p["w"], w0 - lr_t * m / (np.sqrt(v) + 1e-7), rtol=1e-12)
    assert np.allclose(p["ln.g"], 1.0 - lr_t * 0.01 / (np.sqrt(0.001 * 0.01) + 1e-7), rtol=1e-12)  # no decay


def test_metrics_and_shard_rule():
    cm = oo.confusion_matrix([0, 1, 2, 2, 1], [0, 2, 2, 2, 1], 3)
    assert cm.tolist() == [[1, 0, 0], [0, 1, 1], [0, 0, 2]]
    # per-class F1: c0 = 1, c1: p=1 r=.5 -> 2/3, c2: p=2/3 r=1 -> .8
    assert abs(oo.macro_f1(cm) - (1 + 2 / 3 + 0.8) / 3) < 1e-12
    assert oo.macro_f1(np.zeros((3, 3), np.int32)) == 0.0      # divide_no_nan
    assert oo.shard_indices(10, 4, 1) == [1, 5, 9]              # polus/data.py:94-96
