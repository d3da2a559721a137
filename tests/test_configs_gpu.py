"""GPU: every BASELINE.json config at its OWN workload, each anchored to the float64 NumPy oracle
where the oracle is affordable (the forward pass of a few samples of the batch is: a sample's logits
depend on that sample alone) and to size-independent properties otherwise.

  configs[1]  BioBERT-base token classification  S=128  B=32            test_c2_*
  configs[2]  BioBERT-base  S=256  B=64 per GPU                          tests/test_fullsize_gpu.py (+ oracle anchor here)
  configs[3]  PubMedBERT-large dual encoder  S=512  bf16                 test_c4_*
  configs[4]  BioBERT-base  S=512  grad-accum x4  bf16                   test_c5_*
  full depth  BERT-base L=12: f32 AND bf16 engines vs the oracle, every gradient, per element

Tolerances.  f32 engine (exact-f32 MFMA; summation order is the only difference from the oracle):
loss 2e-5, logits 1e-4 of max|ref|, gradients per element |a-r| <= 1.5e-5 |r| + 1.5e-5 rms(r) at full depth.
bf16 engine (bf16 activations and weight shadows, f32 accumulation): loss 2e-2, logits 3e-2 of
max|ref|, gradients cosine > 0.995 and norm within 3 % per tensor, per element
|a-r| <= 0.09 |r| + 0.18 rms(r) (rms over the non-zero entries of r).
The per-element bounds of the full-depth test are twice the worst ratio measured on MI355X in round 4 (the test prints the
three worst tensors: f32 0.012 x (5e-4, 5e-4) at layer0.ffn2.w, bf16 0.43 x (0.1, 0.2) at emb.word)."""
import numpy as np
import pytest
import torch

from oracle import bert as ob
from oracle import losses as ol
from tests.util import assert_close, assert_close_elem, cosine, elem_err, host

pytestmark = pytest.mark.gpu
VOCAB, C = 28996, 4


def synth(B, S, seed, min_len=None):
    import bench
    return bench.synth_batch(B, S, seed)


def dev_batch(ids, mask, tt, labels):
    d = "cuda"
    return ({"input_ids": torch.from_numpy(ids).to(d), "attention_mask": torch.from_numpy(mask).to(d),
             "token_type_ids": torch.from_numpy(tt).to(d)}, torch.from_numpy(labels).to(d))


_ORACLE = {}


def oracle_setup(large=False):
    """Seeded float64 parameters (oracle.golden_setup) of BERT-base / BERT-large, built once per session."""
    if large not in _ORACLE:
        cfg = ob.BertConfig(VOCAB, 1024, 24, 16, 4096, 512, 2) if large else ob.BertConfig(VOCAB, 768, 12, 12, 3072, 512, 2)
        _ORACLE[large] = (cfg,) + tuple(ob.golden_setup(cfg, C))
    return _ORACLE[large]


def build(ocfg, params, hw, hb, mode, num_labels=C):
    from polus_amd.models import BertConfig, BertModel
    cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                     ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size)
    m = BertModel(cfg, compute_dtype=mode, num_labels=num_labels)
    m.load_numpy_params(params, hw, hb)
    return m


# ------------------------------------------------------------------------------ full-depth anchor
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_depth_bert_base_matches_oracle(mode):
    """BERT-base, all 12 layers, B=2 (one ragged) S=128: loss, logits and EVERY parameter gradient of
    both engines against the float64 oracle -- the f32 engine per element, not per tensor."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    ocfg, params, hw, hb = oracle_setup()
    ids, mask, tt, labels = synth(2, 128, 5)
    mask[1, 77:] = 0; ids[1, 77:] = 0; labels[1, 77:] = 0
    ref_loss, ref_logits, cache = ob.token_classifier_fwd(params, ocfg, hw, hb, ids, mask, labels, tt)
    og = ob.token_classifier_bwd(params, ocfg, hw, cache)
    model = build(ocfg, params, hw, hb, mode)
    loss_fn = SparseCategoricalCrossentropy(grad_dtype=model.compute_dtype)
    x, y = dev_batch(ids, mask, tt, labels)
    logits = model(**x, training=True)
    loss = float(loss_fn(y, logits))
    model.backward(loss_fn.backward())
    torch.cuda.synchronize()
    f32 = mode == "f32"
    assert abs(loss - ref_loss) < (2e-5 if f32 else 2e-2), (loss, ref_loss)
    assert_close(host(logits), ref_logits, 1e-4 if f32 else 3e-2, "logits")
    worst = {}
    for v in model.trainable_weights:
        got, ref = host(v.grad), og[v.name]
        if f32:
            worst[v.name] = elem_err(got, ref, 1.5e-5, 1.5e-5)
        else:
            n = np.linalg.norm(ref)
            if n > 1e-12 and v.size >= 256:
                assert cosine(got, ref) > 0.995 and abs(np.linalg.norm(got) / n - 1) < 0.03, (v.name, cosine(got, ref), np.linalg.norm(got) / n)
            worst[v.name] = elem_err(got, ref, 0.09, 0.18)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"full depth, {mode}: worst per-element gradient error / bound: " + ", ".join(f"{k} {e:.3f}" for k, e in top))
    bad = {k: round(e, 3) for k, e in worst.items() if not e <= 1.0}
    assert not bad, f"{mode}: per-element gradient error beyond tolerance: {bad}"


# ------------------------------------------------------------------------------ configs[1]
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_c2_bert_base_b32_s128_workload(mode):
    """configs[1] at its own size: logits of three samples of the batch against the oracle, then one
    ClassifierTrainer step (AdamW): the loss is the oracle's mean over the batch of 32
    and goes down on the same batch."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.training import ClassifierTrainer
    ocfg, params, hw, hb = oracle_setup()
    B, S = 32, 128
    ids, mask, tt, labels = synth(B, S, 21)
    model = build(ocfg, params, hw, hb, mode)
    x, y = dev_batch(ids, mask, tt, labels)
    logits = host(model(**x, training=False))
    f32 = mode == "f32"
    pick = [0, 17, 31]
    _, ref_logits, _ = ob.token_classifier_fwd(params, ocfg, hw, hb, ids[pick], mask[pick], labels[pick], tt[pick])
    assert_close(logits[pick], ref_logits, 1e-4 if f32 else 3e-2, "logits of samples 0, 17, 31")
    # the batch loss from the engine's own logits through the oracle's loss (mean over all B*S positions)
    ref_loss, _ = ol.sparse_softmax_xent_fwd(logits.reshape(-1, C), labels.reshape(-1))
    trainer = ClassifierTrainer(model, AdamWeightDecay(learning_rate=2e-5, weight_decay_rate=0.01),
                                SparseCategoricalCrossentropy(grad_dtype=model.compute_dtype))
    l0 = float(trainer.train_step(x, y))
    assert abs(l0 - ref_loss) < (2e-5 if f32 else 2e-2), (l0, ref_loss)
    l1 = float(trainer.train_step(x, y))
    l2 = float(trainer.train_step(x, y))
    assert l2 < l0, (l0, l1, l2)


# ------------------------------------------------------------------------------ configs[2] anchor
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_c3_bert_base_b64_s256_logits_match_oracle(mode):
    """configs[2] per-GPU workload (64 x 256), the shape bench.py's headline is quoted on: the logits of two samples of the
    batch against the oracle, for BOTH engines (f32 1e-4, bf16 3e-2 of max|ref|); tests/test_fullsize_gpu.py adds the
    engine-vs-engine and loss@step100 checks at the same size."""
    ocfg, params, hw, hb = oracle_setup()
    ids, mask, tt, labels = synth(64, 256, 11)
    model = build(ocfg, params, hw, hb, mode)
    x, _ = dev_batch(ids, mask, tt, labels)
    logits = host(model(**x, training=False))
    pick = [3, 60]
    _, ref, _ = ob.token_classifier_fwd(params, ocfg, hw, hb, ids[pick], mask[pick], labels[pick], tt[pick])
    assert_close(logits[pick], ref, 1e-4 if mode == "f32" else 3e-2, f"logits of samples 3, 60 ({mode})")


# ------------------------------------------------------------------------------ configs[4]
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_c5_bert_base_s512_grad_accum_x4(mode):
    """configs[4]: S=512, 64 samples as 4 micro-batches of 16 with trainer.grad_accum_steps = 4.
    One sample's logits against the oracle; the accumulated gradient equals the gradient of the whole
    batch of 64; exactly ONE optimizer step is taken, on the 4th micro-step, and it equals the step of
    the whole batch."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.training import ClassifierTrainer
    ocfg, params, hw, hb = oracle_setup()
    B, S, A = 64, 512, 4
    ids, mask, tt, labels = synth(B, S, 31)
    f32 = mode == "f32"
    x, y = dev_batch(ids, mask, tt, labels)

    def trainer_of(accum):
        m = build(ocfg, params, hw, hb, mode)
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=1e-4, weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        t.grad_accum_steps = accum
        return m, t

    m_acc, t_acc = trainer_of(A)
    logits = host(m_acc(**{k: v[:16] for k, v in x.items()}, training=False))
    _, ref, _ = ob.token_classifier_fwd(params, ocfg, hw, hb, ids[5:6], mask[5:6], labels[5:6], tt[5:6])
    assert_close(logits[5:6], ref, 1e-4 if f32 else 3e-2, "logits of sample 5 (S=512)")
    p0 = m_acc.arena.params.clone()
    losses = []
    for k in range(A):
        sl = slice(16 * k, 16 * (k + 1))
        losses.append(float(t_acc.train_step({n: v[sl] for n, v in x.items()}, y[sl])))
        if k < A - 1:
            assert torch.equal(p0, m_acc.arena.params), "parameters moved before the last micro-step"
            assert t_acc.optimizer.iterations == 0
    assert t_acc.optimizer.iterations == 1 and not torch.equal(p0, m_acc.arena.params)
    g_acc = m_acc.arena.grads.clone()

    m_one, t_one = trainer_of(1)
    l_one = float(t_one.train_step(x, y))
    assert t_one.optimizer.iterations == 1
    # mean over 64*512 positions == mean of the four micro-batch means (equal sizes)
    assert abs(np.mean(losses) - l_one) < (1e-5 if f32 else 5e-3), (losses, l_one)
    for v in m_one.arena.vars:
        a = m_one.arena.grads[v.offset:v.offset + v.size].double()
        b = g_acc[v.offset:v.offset + v.size].double() / A
        na = a.norm().item()
        if v.size < 256 or na < 1e-9:
            continue
        assert (a - b).norm().item() < (2e-4 if f32 else 3e-2) * na, (v.name, (a - b).norm().item() / na)
    # and the one optimizer step lands in the same place (AdamW's first step is lr * sign-like: compare the update)
    u_acc = (m_acc.arena.params - p0).double()
    u_one = (m_one.arena.params - p0).double()
    cos = (u_acc @ u_one).item() / (u_acc.norm().item() * u_one.norm().item() + 1e-300)
    assert cos > (0.999 if f32 else 0.90), cos


# ------------------------------------------------------------------------------ configs[3]
def test_c4_bert_large_dual_encoder_s512():
    """configs[3]: BERT-large (L=24, H=1024) dual encoder, S=512, bf16, through
    EfficientDenseRetrievalTrainer (polus/ir/training.py:47-117).  Per engine: the [CLS] states of one query and one
    document against the oracle (f32 1e-4, bf16 5e-2 of max|ref|); the encoder arena bit-unchanged by the step (no
    gradient reaches BERT); the reported loss and the projection gradients against the oracle applied to THAT engine's
    own [CLS] states (projections -> in-batch scores -> softmax CE on the diagonal).  End to end: the bf16 engine's loss
    against the f32 engine's (which the two oracle anchors and its own-state check tie to the oracle at 1e-4).

    Conditioning.  With every sequence starting with the same [CLS] token a random-init BERT-large maps all inputs to
    nearly the same state: the 8 x 8 in-batch loss then hangs on differences the size of the bf16 encoder's 1.2e-2
    rounding noise, and the bf16-vs-f32 gap read +0.03, +0.17 and +0.32 under three builds with identical encoder error.
    So the inputs here differ in their first token (the state at position 0 is then input-specific by far more than the
    noise) and the projections are drawn at sigma = 0.013, which puts the standard deviation of the scores near 2 --
    inside the range where softmax CE responds to them without saturating.  The end-to-end gap is then asserted."""
    from polus_amd.ir.models import DualEncoder
    from polus_amd.ir.training import ContrastiveLoss, EfficientDenseRetrievalTrainer, InBatchDotScores
    from polus_amd.optimizers import Adam
    ocfg, params, _, _ = oracle_setup(large=True)
    B, S, E = 8, 512, 128
    qi, qm, _, _ = synth(B, S, 41)
    di, dm, _, _ = synth(B, S, 42)
    first = np.random.Generator(np.random.PCG64(43)).permutation(np.arange(2000, 2000 + 2 * B)).astype(np.int32)
    qi[:, 0], di[:, 0] = first[:B], first[B:]                # input-specific first tokens (see Conditioning)
    q = {"input_ids": torch.from_numpy(qi).cuda(), "attention_mask": torch.from_numpy(qm).cuda()}
    d = {"input_ids": torch.from_numpy(di).cuda(), "attention_mask": torch.from_numpy(dm).cuda()}
    ref_cls = ob.bert_fwd(params, ocfg, qi[:1], qm[:1])[1]
    ref_cls_d = ob.bert_fwd(params, ocfg, di[3:4], dm[3:4])[1]
    pr = np.random.Generator(np.random.PCG64(44))
    proj = [pr.standard_normal((E, ocfg.hidden_size)) * 0.013, np.zeros(E), pr.standard_normal((E, ocfg.hidden_size)) * 0.013, np.zeros(E)]
    losses, rel_hidden = {}, {}
    for mode in ("f32", "bf16"):
        enc = build(ocfg, params, None, None, mode, num_labels=None)
        cls = host(enc(**{k: v[:1] for k, v in q.items()}, training=False).pooler_output)
        assert_close(cls, ref_cls, 1e-4 if mode == "f32" else 5e-2, f"[CLS] of query 0 ({mode})")
        cls_d = host(enc(**{k: v[3:4] for k, v in d.items()}, training=False).pooler_output)
        assert_close(cls_d, ref_cls_d, 1e-4 if mode == "f32" else 5e-2, f"[CLS] of document 3 ({mode})")
        out_q = enc(**q, training=False)
        hid = out_q.last_hidden_state.float().clone()      # the output lives in the model's scratch: copy before the next forward
        hq = host(out_q.pooler_output).astype(np.float64)
        if mode == "f32":
            hid32 = hid
        rel_hidden[mode] = float((hid - hid32).norm() / hid32.norm())
        del out_q
        hd = host(enc(**d, training=False).pooler_output).astype(np.float64)
        model = DualEncoder(enc, projection_dim=E, compute_dtype=mode)
        for v, a in zip(model.trainable_weights, proj):
            v.assign(a.astype(np.float32))
        w = {v.name: v.numpy().astype(np.float64) for v in model.trainable_weights}
        n = [v.name for v in model.trainable_weights]
        before = enc.arena.params.clone()
        trainer = EfficientDenseRetrievalTrainer(model, InBatchDotScores(), optimizer=Adam(1e-3), loss=ContrastiveLoss())
        losses[mode] = float(trainer.train_step(q, d))
        torch.cuda.synchronize()
        assert torch.equal(before, enc.arena.params), "the frozen encoder moved"
        got = {v.name: host(v.grad) for v in model.trainable_weights}
        # oracle on this engine's own [CLS] states
        wq, bq, wd, bd = (w[k] for k in n)
        pq, pd_ = hq @ wq.T + bq, hd @ wd.T + bd
        loss_ref, ds = ol.sparse_softmax_xent_fwd(pq @ pd_.T, np.arange(B))
        assert abs(losses[mode] - loss_ref) < (1e-4 if mode == "f32" else 5e-2) * max(1.0, abs(loss_ref)), (mode, losses[mode], loss_ref)
        dq, dd = ds @ pd_, ds.T @ pq
        tol = 5e-4 if mode == "f32" else 8e-2
        assert_close(got[n[0]], dq.T @ hq, tol, f"query projection dW ({mode})")
        assert_close(got[n[1]], dq.sum(0), tol, f"query projection db ({mode})")
        assert_close(got[n[2]], dd.T @ hd, tol, f"document projection dW ({mode})")
        # d loss / d (document bias) is zero analytically (every row of softmax - onehot sums to zero): absolute bound
        assert np.abs(got[n[3]]).max() <= tol * np.abs(dq.sum(0)).max(), (mode, np.abs(got[n[3]]).max(), np.abs(dq.sum(0)).max())
        del trainer, model, enc
        torch.cuda.empty_cache()
    assert rel_hidden["bf16"] < 2e-2, rel_hidden
    print(f"c4 first-step loss: f32 {losses['f32']:.4f}  bf16 {losses['bf16']:.4f}  log(B) {np.log(B):.4f}")
    assert abs(losses["f32"] - np.log(B)) > 0.05, "scores too small: the in-batch loss would not see the encoders at all"
    assert abs(losses["bf16"] - losses["f32"]) < 5e-2 * max(1.0, abs(losses["f32"])), losses


def test_c4_dual_encoder_at_the_benched_size():
    """configs[3] at the size `bench.py --config c4` runs: 64 query-document pairs x 512 tokens, BERT-large.  The float64
    oracle is out of reach here (64 x 512 x 24 layers), so: size-independent properties -- the bf16 engine's [CLS] states of all 64
    queries against the f32 engine's (the oracle anchors of the test above tie that engine to the oracle at 1e-4), the
    first-step losses of the two engines, and the encoder arena bit-unchanged by the training step (nothing is differentiated
    through BERT, polus/ir/training.py:47-117)."""
    from polus_amd.ir.models import DualEncoder
    from polus_amd.ir.training import ContrastiveLoss, EfficientDenseRetrievalTrainer, InBatchDotScores
    from polus_amd.optimizers import Adam
    ocfg, params, _, _ = oracle_setup(large=True)
    B, S, E = 64, 512, 128
    qi, qm, _, _ = synth(B, S, 51)
    di, dm, _, _ = synth(B, S, 52)
    first = np.random.Generator(np.random.PCG64(53)).permutation(np.arange(2000, 2000 + 2 * B)).astype(np.int32)
    qi[:, 0], di[:, 0] = first[:B], first[B:]                # input-specific first tokens (see the test above, Conditioning)
    q = {"input_ids": torch.from_numpy(qi).cuda(), "attention_mask": torch.from_numpy(qm).cuda()}
    d = {"input_ids": torch.from_numpy(di).cuda(), "attention_mask": torch.from_numpy(dm).cuda()}
    pr = np.random.Generator(np.random.PCG64(54))
    proj = [pr.standard_normal((E, ocfg.hidden_size)) * 0.013, np.zeros(E), pr.standard_normal((E, ocfg.hidden_size)) * 0.013, np.zeros(E)]
    cls, losses = {}, {}
    for mode in ("f32", "bf16"):
        enc = build(ocfg, params, None, None, mode, num_labels=None)
        cls[mode] = host(enc(**q, training=False).pooler_output).astype(np.float64)
        model = DualEncoder(enc, projection_dim=E, compute_dtype=mode)
        for v, a in zip(model.trainable_weights, proj):
            v.assign(a.astype(np.float32))
        before = enc.arena.params.clone()
        trainer = EfficientDenseRetrievalTrainer(model, InBatchDotScores(), optimizer=Adam(1e-3), loss=ContrastiveLoss())
        losses[mode] = float(trainer.train_step(q, d))
        torch.cuda.synchronize()
        assert torch.equal(before, enc.arena.params), f"the frozen encoder moved ({mode})"
        del trainer, model, enc, before
        torch.cuda.empty_cache()
    assert_close(cls["bf16"], cls["f32"], 5e-2, "[CLS] states of the 64 queries, bf16 engine against f32 engine")
    print(f"c4 at 64 x 512: first-step loss f32 {losses['f32']:.4f}  bf16 {losses['bf16']:.4f}  log(B) {np.log(B):.4f}")
    assert np.isfinite(losses["f32"]) and abs(losses["bf16"] - losses["f32"]) < 5e-2 * max(1.0, abs(losses["f32"])), losses
