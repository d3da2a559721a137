"""GPU: the BASELINE.json headline shape itself (BERT-base L=12, 64 x 256 tokens), where the NumPy
oracle would need minutes per step.  Parity here rests on size-independent properties:

* the bf16 engine against the f32 engine (exact-f32 MFMA, the engine the golden vectors and the
  100-step trajectory pin to the oracle at small sizes) on identical weights and batch: loss, logits,
  gradient direction and norm per tensor class;
* two micro-batches with accumulation == one batch (linearity of the backward pass);
* the whole training step is bitwise reproducible with dropout on (counter-based masks, fixed-order
  split-K / LayerNorm reductions, owner-wave embedding scatter).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S, C = 64, 256, 4


def _batch(seed):
    import bench
    ids, mask, tt, labels = bench.synth_batch(B, S, seed)
    dev = "cuda"
    return ({"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev),
             "token_type_ids": torch.from_numpy(tt).to(dev)}, torch.from_numpy(labels).to(dev))


def _model(dtype, p=0.0):
    import bench
    from polus_amd.models import BertConfig, BertModel
    cfg = BertConfig(vocab_size=bench.VOCAB, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, max_position_embeddings=512,
                     hidden_dropout_prob=p, attention_probs_dropout_prob=p)
    return BertModel(cfg, compute_dtype=dtype, num_labels=C, seed=1234)


def _fwd_bwd(model, x, y, accumulate=False):
    from polus_amd.losses import SparseCategoricalCrossentropy
    loss_fn = SparseCategoricalCrossentropy(grad_dtype=model.compute_dtype)
    logits = model(**x, training=True)
    loss = float(loss_fn(y, logits))
    model.backward(loss_fn.backward(), accumulate=accumulate)
    torch.cuda.synchronize()
    return loss, logits.float().cpu().numpy()


def test_bf16_engine_tracks_f32_engine_at_full_size():
    x, y = _batch(11)
    m32, m16 = _model("f32"), _model("bf16")
    assert torch.equal(m32.arena.params, m16.arena.params), "same seed, same initial weights"
    l32, z32 = _fwd_bwd(m32, x, y)
    l16, z16 = _fwd_bwd(m16, x, y)
    assert abs(l32 - l16) < 2e-2, (l32, l16)                       # tolerance of tests/test_model_gpu.py (bf16)
    assert np.abs(z32 - z16).max() < 3e-2 * max(1.0, np.abs(z32).max())
    for v32, v16 in zip(m32.arena.vars, m16.arena.vars):
        if v32.size < 1000:
            continue
        g32, g16 = v32.grad.double().flatten(), v16.grad.double().flatten()
        n32, n16 = g32.norm().item(), g16.norm().item()
        if n32 < 1e-7:
            continue
        cos = (g32 @ g16).item() / (n32 * n16 + 1e-30)
        assert cos > 0.99 and abs(n16 / n32 - 1) < 0.05, (v32.name, cos, n16 / n32)


def test_accumulation_equals_whole_batch_at_full_size():
    x, y = _batch(12)
    m = _model("bf16")
    _fwd_bwd(m, x, y)
    whole = m.arena.grads.clone()
    half = lambda a, lo: {k: v[lo:lo + B // 2] for k, v in a.items()}
    _fwd_bwd(m, half(x, 0), y[:B // 2])
    _fwd_bwd(m, half(x, B // 2), y[B // 2:], accumulate=True)
    two = m.arena.grads * 0.5          # each half-batch loss is a mean over half the tokens
    for v in m.arena.vars:
        a, b = whole[v.offset:v.offset + v.size].double(), two[v.offset:v.offset + v.size].double()
        na = a.norm().item()
        if v.size < 1000 or na < 1e-7:
            continue
        assert (a - b).norm().item() < 3e-2 * na, (v.name, (a - b).norm().item() / na)


def test_training_step_is_bitwise_reproducible_at_full_size():
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    runs = []
    for _ in range(2):
        m = _model("bf16", p=0.1)
        m.deterministic = True
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=warmup_scheduler(100, 5e-5), weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        losses = [float(t.train_step(*_batch(20 + k))) for k in range(3)]
        torch.cuda.synchronize()
        runs.append((losses, m.arena.params.clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]), "parameters differ after three identical steps"
    assert runs[0][0][-1] < runs[0][0][0], "and the loss goes down"


LOSS100 = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden",
                                     "loss100_bert_base_b64_s256.npz")


@pytest.mark.parametrize("dtype,tol1,tol100,tolmax", [("f32", 1e-5, 1e-3, 2e-3), ("bf16", 5e-3, 5e-3, 1e-1)])
def test_loss_at_step100_matches_the_oracle_curve_at_the_headline_shape(dtype, tol1, tol100, tolmax):
    """BASELINE.json: "loss within 1e-3 of the reference at step 100" -- at the headline shape itself.  The committed
    fixture is the 100-step loss curve of the ORACLE (oracle/bert_torch.py in float64; tests/golden/make_loss100.py) for
    BERT-base, 64 x 256 tokens, dropout 0, the bench's AdamW / warm-up schedule and its 8 recurring batches.  The engines
    start from bit-identical weights (asserted) and must follow the curve: f32 within 1e-5 at step 1 and 1e-3 at step 100
    (and never further than 2e-3 on the way); the bf16 engine within 5e-3 at steps 1 and 100 (bf16 rounding of activations
    and weight shadows; bench.py reports its actual gap, 5e-4 at step 100) and within 0.1 on the way -- during the steep
    descent of steps 2-15 (1.93 -> 1.1) two trajectories a rounding apart pass the same loss a fraction of a step apart,
    which reads as a transient gap of a few 1e-2 that closes again."""
    import os
    import types
    import bench
    if not os.path.exists(LOSS100):
        pytest.skip("tests/golden/loss100_bert_base_b64_s256.npz not generated yet (tests/golden/make_loss100.py, ~2 h of CPU)")
    ref = np.load(LOSS100, allow_pickle=False)["loss"]
    assert len(ref) >= 100
    args = types.SimpleNamespace(geom=(768, 12, 3072, 12), kind="ner", accum=1, batch=B, seq=S)
    model, trainer = bench.build_trainer(args, dtype, 0.0, 100)
    model.deterministic = True
    from tests.golden.make_loss100 import initial_weights
    _, init = initial_weights()
    for v in model.arena.vars:
        assert np.array_equal(v.numpy(), init[v.name].reshape(v.shape)), f"initial {v.name} differs from the fixture's recipe"
    batches = bench.device_batches(args, 0, model.arena.device, n=8, seed0=100)
    curve = np.asarray([float(trainer.train_step(*batches[s % 8])) for s in range(100)])
    gap = np.abs(curve - ref[:100])
    print(f"loss@100 {dtype}: engine {curve[99]:.6f} oracle {ref[99]:.6f}  |gap| step1 {gap[0]:.2e} step100 {gap[99]:.2e} "
          f"max {gap.max():.2e} at step {int(gap.argmax()) + 1}, mean over the last 50 steps {gap[50:].mean():.2e}")
    assert gap[0] <= tol1 and gap[99] <= tol100 and gap.max() <= tolmax, (gap[0], gap[99], gap.max())
    assert gap[50:].mean() <= 2 * tol100
    assert curve[99] < curve[0] - 0.1, "the model should have learned something in 100 steps"
