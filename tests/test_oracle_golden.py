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


def test_adam_pinned_to_torch_optim():
    """oracle.optim.Adam (the Keras form: p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps), eps OUTSIDE the
    bias-corrected root) against torch.optim.Adam / AdamW, an independent implementation.  torch divides
    by sqrt(v/(1-b2^t)) + eps_t; the two are the same update when eps_t = eps/sqrt(1-b2^t):
        lr/(1-b1^t) * m / (sqrt(v)/sqrt(c2) + eps_t) = lr*sqrt(c2)/(1-b1^t) * m / (sqrt(v) + eps_t*sqrt(c2)).
    So (a) with that per-step eps_t the trajectories agree to rounding -- this pins the moment updates,
    both bias corrections, the decoupled decay (AdamW: p *= 1 - lr*wd with the UNcorrected lr, as HF
    AdamWeightDecay) and the epsilon placement; (b) with the same constant eps on both sides they differ by
    at most |dp| * eps*(1/sqrt(c2)-1)/sqrt(v), checked as a bound."""
    import torch
    r = np.random.Generator(np.random.PCG64(5))
    shapes = {"w": (7, 5), "ln.g": (5,), "b": (7,)}
    for wd in (0.0, 0.01):
        p = {k: r.standard_normal(s) for k, s in shapes.items()}
        tp = {k: torch.tensor(v.copy(), dtype=torch.float64, requires_grad=True) for k, v in p.items()}
        nod = ["ln.g", "b"]
        opt = oo.Adam(lr=3e-3, eps=1e-7, weight_decay=wd, no_decay=nod)
        groups = [{"params": [tp["w"]], "weight_decay": wd}, {"params": [tp[k] for k in nod], "weight_decay": 0.0}]
        topt = torch.optim.AdamW(groups, lr=3e-3, betas=(0.9, 0.999), eps=1e-7)
        for t in range(1, 13):
            g = {k: r.standard_normal(s) * 10.0 ** r.integers(-6, 1) for k, s in shapes.items()}
            for k in tp:
                tp[k].grad = torch.tensor(g[k].copy(), dtype=torch.float64)
            for grp in topt.param_groups:
                grp["eps"] = 1e-7 / np.sqrt(1.0 - 0.999 ** t)
            opt.step(p, g)
            topt.step()
            for k in p:
                assert np.allclose(p[k], tp[k].detach().numpy(), rtol=1e-12, atol=1e-15), (wd, t, k)
    # (b) same constant eps on both sides: bounded difference after one step from identical state
    p = {"w": r.standard_normal((50,))}
    g = {"w": r.standard_normal((50,)) * 1e-3}
    tw = torch.tensor(p["w"].copy(), dtype=torch.float64, requires_grad=True)
    tw.grad = torch.tensor(g["w"].copy(), dtype=torch.float64)
    oo.Adam(lr=1e-3, eps=1e-7).step(p, g)
    torch.optim.Adam([tw], lr=1e-3, eps=1e-7).step()
    c2 = 1.0 - 0.999
    step = 1e-3                                   # |dp| <= lr at t = 1
    bound = step * 1e-7 * (1.0 / np.sqrt(c2) - 1.0) / np.sqrt(0.001 * g["w"] ** 2)
    assert np.all(np.abs(p["w"] - tw.detach().numpy()) <= bound * 1.0001 + 1e-18)


def test_schedule_table_against_reference_closed_form():
    """polus/schedulers.py:10-23 = WarmUp(power 1) over PolynomialDecay(power 1, end 1e-7), written out
    independently: every integer step of a run and past its end."""
    for N, pct, lr in ((100, 0.1, 1e-3), (1000, 0.1, 5e-5), (37, 0.25, 2e-4)):
        W = int(N * pct)
        D = N - W
        for t in range(0, 2 * N + 3):
            if t < W:
                ref = lr * (t / W) ** 1.0
            else:
                s = min(t - W, D)
                ref = (lr - 1e-7) * (1.0 - s / D) ** 1.0 + 1e-7
            assert abs(oo.warmup_linear_lr(t, N, lr, pct) - ref) <= 1e-18 + 1e-15 * ref, (N, t)


def test_schedule_pinned_to_hf_polynomial_decay_with_warmup():
    """An INDEPENDENT implementation of the same schedule: transformers.optimization.get_polynomial_decay_schedule_with_warmup
    (power 1, lr_end 1e-7) is the PyTorch twin of HF-TF's WarmUp over PolynomialDecay that polus/schedulers.py:10-23
    builds.  The oracle's closed form and the product's WarmUpLinearDecay must follow it at every step of a run and past
    its end."""
    import torch
    from transformers.optimization import get_polynomial_decay_schedule_with_warmup
    from polus_amd.optimizers import WarmUpLinearDecay
    for N, pct, lr in ((100, 0.1, 1e-3), (1000, 0.1, 5e-5), (37, 0.25, 2e-4)):
        W = int(N * pct)
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
        sched = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=W, num_training_steps=N, lr_end=1e-7, power=1.0)
        mine = WarmUpLinearDecay(N, lr, pct)
        for t in range(0, N + 20):
            ref = sched.get_last_lr()[0]
            assert abs(oo.warmup_linear_lr(t, N, lr, pct) - ref) <= 1e-12 * lr, (N, t, ref)
            assert abs(mine(t) - ref) <= 1e-12 * lr, (N, t, ref)
            opt.step()
            sched.step()


def test_torch_cpu_port_matches_numpy_oracle():
    """oracle/bert_torch.py (the cpu_baseline leg of bench.py) computes the same step as the NumPy
    oracle: loss, logits, every gradient (autograd vs the explicit backward) and three AdamW steps."""
    import torch
    from oracle import bert_torch as bt
    cfg = ob.BertConfig(97, 128, 2, 2, 256, 64, 2)
    params, hw, hb = ob.golden_setup(cfg, 4)
    r = np.random.Generator(np.random.PCG64(9))
    ids = r.integers(1, 97, size=(3, 24)).astype(np.int32)
    mask = (np.arange(24)[None] < np.array([24, 9, 17])[:, None]).astype(np.int32)
    tt = (r.integers(0, 2, size=(3, 24)) * mask).astype(np.int32)
    labels = (r.integers(0, 4, size=(3, 24)) * mask).astype(np.int32)
    allp = dict(params); allp["head.w"] = hw; allp["head.b"] = hb
    tp = bt.to_torch(allp, torch.float64)
    loss, logits, cache = ob.token_classifier_fwd(allp, cfg, allp["head.w"], allp["head.b"], ids, mask, labels, tt)
    tl, tlogits = bt.token_classifier_loss(tp, cfg, ids, mask, labels, tt)
    assert abs(float(tl.detach()) - loss) < 1e-12
    assert np.abs(tlogits.detach().numpy() - logits).max() < 1e-11
    tl.backward()
    grads = ob.token_classifier_bwd(allp, cfg, allp["head.w"], cache)
    for k, g in grads.items():
        assert np.abs(tp[k].grad.numpy() - g).max() <= 1e-10 * max(1.0, np.abs(g).max()), k
    for k in tp:
        tp[k].grad = None
    nod = [k for k in allp if oo.is_no_decay(k)]
    o_np = oo.Adam(lr=lambda t: oo.warmup_linear_lr(t, 10, 1e-3), weight_decay=0.01, no_decay=nod)
    o_t = bt.Adam(lr=lambda t: oo.warmup_linear_lr(t, 10, 1e-3), weight_decay=0.01, no_decay=nod)
    for s in range(3):
        l_np, _, cache = ob.token_classifier_fwd(allp, cfg, allp["head.w"], allp["head.b"], ids, mask, labels, tt)
        o_np.step(allp, ob.token_classifier_bwd(allp, cfg, allp["head.w"], cache))
        l_t = bt.train_step(tp, cfg, o_t, ids, mask, labels, tt)
        assert abs(l_np - l_t) < 1e-10, (s, l_np, l_t)
    for k in allp:
        assert np.abs(tp[k].detach().numpy() - allp[k]).max() < 1e-10, k


def test_pooler_matches_hf_twin():
    """oracle.bert.pooler_fwd / pooler_bwd against transformers' BertPooler (float64, autograd)."""
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    from transformers.models.bert.modeling_bert import BertPooler
    hc = tr.BertConfig(hidden_size=64, num_attention_heads=1, num_hidden_layers=1, intermediate_size=64, vocab_size=10)
    torch.manual_seed(1)
    pool = BertPooler(hc).double()
    r = np.random.Generator(np.random.PCG64(4))
    h = r.standard_normal((3, 7, 64))
    w, b = pool.dense.weight.detach().numpy(), pool.dense.bias.detach().numpy()
    ht = torch.tensor(h, requires_grad=True)
    out = pool(ht)
    pooled, cache = ob.pooler_fwd(h, w, b)
    assert np.abs(pooled - out.detach().numpy()).max() < 1e-13
    dp = r.standard_normal((3, 64))
    out.backward(torch.tensor(dp))
    dlast, dw, db = ob.pooler_bwd(dp, w, cache, 7)
    assert np.abs(dlast - ht.grad.numpy()).max() < 1e-12
    assert np.abs(dw - pool.dense.weight.grad.numpy()).max() < 1e-12 and np.abs(db - pool.dense.bias.grad.numpy()).max() < 1e-12
