"""GPU: the pieces either side of the training step through the
HIP engine, each against the float64 oracle or an exact round trip:

* explicit-negatives branch of EfficientDenseRetrievalTrainer (polus/ir/training.py:94-117);
* HF BertPooler for the unsplit model (tanh(W h[:,0] + b)), forward and backward;
* `.cfg` / `.init` / weights: SavableModel.save -> load_model rebuilds an identical model (Sequential and BERT + head);
* a local safetensors checkpoint loaded into the HIP BertModel reproduces the golden logits;
* optimizer / step state round trip: a resumed run continues bit for bit;
* Dataset.prefetch hands over device tensors in order; the GPU confusion matrix equals the host one."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bert as ob
from oracle import losses as ol
from oracle import optim as oo
from tests.util import assert_close, dev, host
from tests.test_model_gpu import GOLD, build_model, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_ir_explicit_negatives_step_matches_oracle(mode):
    """k = 3 negative documents per query: the positives and negatives go through document_projection as one
    [(k+1)B, H] call; the document projection's dW must contain the negatives' contribution."""
    from polus_amd.ir.models import DualEncoder
    from polus_amd.ir.training import ContrastiveLoss, EfficientDenseRetrievalTrainer, InBatchDotScores
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import Adam
    g, ocfg, params, _, _ = load_case("bert_small_b3_s48")
    cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                     ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size)
    enc = BertModel(cfg, compute_dtype=mode); enc.load_numpy_params(params)
    B, S, E, K = 64, 16, 128, 3
    r = np.random.Generator(np.random.PCG64(13))
    mk = lambda *shape: r.integers(1, ocfg.vocab_size, size=shape).astype(np.int32)
    q = {"input_ids": mk(B, S), "attention_mask": np.ones((B, S), np.int32)}
    d = {"input_ids": mk(B, S), "attention_mask": np.ones((B, S), np.int32)}
    nmask = np.ones((B, K, S), np.int32); nmask[:, :, 11:] = 0
    n = {"input_ids": mk(B, K, S) * nmask, "attention_mask": nmask}
    model = DualEncoder(enc, projection_dim=E, compute_dtype=mode)
    before = enc.arena.params.clone()
    w = {v.name: v.numpy().astype(np.float64) for v in model.trainable_weights}
    trainer = EfficientDenseRetrievalTrainer(model, InBatchDotScores(), optimizer=Adam(1e-3), loss=ContrastiveLoss())
    loss = float(trainer.train_step(q, d, n))
    assert trainer.k_negatives == K and torch.equal(before, enc.arena.params)
    # oracle
    hq = ob.bert_fwd(params, ocfg, q["input_ids"], q["attention_mask"])[1]
    hd = ob.bert_fwd(params, ocfg, d["input_ids"], d["attention_mask"])[1]
    hn = [ob.bert_fwd(params, ocfg, n["input_ids"][:, i], n["attention_mask"][:, i])[1] for i in range(K)]
    names = [v.name for v in model.trainable_weights]
    wq, bq, wd, bd = (w[k] for k in names)
    docs = np.concatenate([hd] + hn, 0)                       # [(K+1)B, H]
    pq, pdocs = hq @ wq.T + bq, docs @ wd.T + bd
    scores = pq @ pdocs.T                                     # [B, (K+1)B]
    loss_ref, ds = ol.sparse_softmax_xent_fwd(scores, np.arange(B))
    assert abs(loss - loss_ref) < (1e-4 if mode == "f32" else 5e-2) * max(1.0, abs(loss_ref)), (loss, loss_ref)
    dq, ddocs = ds @ pdocs, ds.T @ pq
    got = {v.name: host(v.grad) for v in model.trainable_weights}
    tol = 5e-4 if mode == "f32" else 8e-2
    assert_close(got[names[0]], dq.T @ hq, tol, "query projection dW")
    assert_close(got[names[2]], ddocs.T @ docs, tol, "document projection dW (positives + negatives)")
    # softmax-CE gradient rows sum to zero, so db = sum_j dDocs[j] = sum_b pq[b] * (sum_j dS[b,j]) vanishes: absolute check
    assert np.abs(got[names[3]]).max() < (1e-5 if mode == "f32" else 2e-2) * np.abs(ddocs).sum(0).max() + 1e-6, "document projection db"
    only_pos = ddocs[:B].T @ hd
    assert np.abs(got[names[2]] - only_pos).max() > 10 * tol * np.abs(only_pos).max(), "negatives contributed nothing"
    # and the step moved the projections
    assert not np.array_equal(model.trainable_weights[2].numpy().astype(np.float64), wd)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_hf_pooler_forward_backward(mode):
    """BertModel(add_pooling_layer=True): pooler_output = tanh(W h[:,0] + b); gradients from a loss on
    BOTH outputs (last_hidden_state and pooler_output) against the oracle."""
    from polus_amd.models import BertConfig, BertModel
    g, ocfg, params, _, _ = load_case("bert_small_b2_s16")
    cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                     ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size)
    m = BertModel(cfg, compute_dtype=mode, add_pooling_layer=True)
    r = np.random.Generator(np.random.PCG64(21))
    H = ocfg.hidden_size
    pw, pb = r.standard_normal((H, H)) * 0.05, r.standard_normal(H) * 0.05
    m.load_numpy_params(dict(params, **{"pooler.w": pw, "pooler.b": pb}))
    assert [v.name for v in m.trainable_weights][-2:] == ["pooler.w", "pooler.b"]
    out = m(input_ids=g["ids"], attention_mask=g["mask"], token_type_ids=g["token_type"], training=True)
    last, _, cache = ob.bert_fwd(params, ocfg, g["ids"], g["mask"], g["token_type"])
    pooled, pc = ob.pooler_fwd(last, pw, pb)
    f32 = mode == "f32"
    assert_close(host(out.pooler_output), pooled, 1e-4 if f32 else 3e-2, "pooler_output")
    assert_close(host(out.last_hidden_state), last, 1e-4 if f32 else 3e-2, "last_hidden_state")
    B, S = g["ids"].shape
    dlast, dpool = r.standard_normal((B, S, H)) * 0.1, r.standard_normal((B, H)) * 0.1
    keep = dlast.copy()
    dl_t = dev(dlast, m.compute_dtype)
    m.backward(dl_t, dpooled=dev(dpool, m.compute_dtype))
    torch.cuda.synchronize()
    assert np.array_equal(host(dl_t), host(dev(keep, m.compute_dtype))), "backward wrote into the caller's gradient"
    from tests.util import rounded
    dl_r, dp_r = rounded(dlast, m.compute_dtype), rounded(dpool, m.compute_dtype)
    dl2, dw, db = ob.pooler_bwd(dp_r, pw, pc, S)
    og = ob.bert_bwd(dl_r + dl2, params, ocfg, cache)
    tol = 3e-4 if f32 else 6e-2
    got = {v.name: host(v.grad) for v in m.trainable_weights}
    assert_close(got["pooler.w"], dw, tol, "pooler dW")
    assert_close(got["pooler.b"], db, tol, "pooler db")
    for k in ("layer1.ffn2.w", "layer0.qkv.w", "emb.pos", "layer1.ln2.g"):
        assert_close(got[k], og[k], tol, k)


def test_save_cfg_init_and_load_model_round_trip(tmp_path):
    """polus/models.py:18-50,60-82,112-133: <name>.cfg + <name>.init + weights; load_model rebuilds through the
    @from_config builder named in the file."""
    import sys
    from polus_amd.models import load_model
    from polus_amd.ner.models import baselineNER_MLP_CRF
    import polus_amd.ner.models as ner_models
    m = baselineNER_MLP_CRF(model={"sequence_length": 12, "output_classes": 3, "hidden_space": 32})
    assert m.name == "baselineNER_MLP_CRF" and m.savable_config["func_name"] == "baselineNER_MLP_CRF"
    assert m.savable_config["model"]["hidden_space"] == 32
    r = np.random.Generator(np.random.PCG64(3))
    x = r.standard_normal((2, 12, 768)).astype(np.float32)
    m.init_from_data(x)
    for v in m.trainable_weights:
        v.assign(r.standard_normal(v.shape).astype(np.float32) * 0.1)
    path = m.save(base_path=str(tmp_path), extension="_best")
    assert all(os.path.exists(path + e) for e in (".cfg", ".init.npz", ".npz"))
    cfg = json.load(open(path + ".cfg"))
    assert cfg["func_name"] == "baselineNER_MLP_CRF" and cfg["model"]["output_classes"] == 3
    m2 = load_model(path + ".cfg", external_module=ner_models)
    assert [v.name for v in m2.trainable_weights] == [v.name for v in m.trainable_weights]
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    assert torch.equal(m(x, training=True), m2(x, training=True))
    m3 = load_model(path + ".cfg", change_config={"hidden_space": 32}, external_module=ner_models)
    assert m3.savable_config["model"]["hidden_space"] == 32


def test_bert_with_head_saves_and_reloads(tmp_path):
    """A fine-tuned BERT + token head is a SavableModel (SaveModelCallback calls model.save,
    polus/callbacks.py:297-313): save -> load_model -> identical logits."""
    from polus_amd.models import load_model
    g, ocfg, params, hw, hb = load_case("bert_small_b2_s16")
    m = build_model(ocfg, params, hw, hb, "f32")
    m.set_name("ner_bert")
    x = {"input_ids": g["ids"], "attention_mask": g["mask"], "token_type_ids": g["token_type"]}
    ref = m(**x, training=False).clone()
    path = m.save(base_path=str(tmp_path), extension="_e3")
    assert os.path.basename(path) == "ner_bert_e3"
    m2 = load_model(path + ".cfg")
    assert m2.head is not None and m2.config.num_hidden_layers == ocfg.num_hidden_layers
    assert torch.equal(m2(**x, training=False), ref)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_local_safetensors_checkpoint_into_hip_model(tmp_path, mode):
    """n2: the golden case's weights written as a local HF directory (config.json + model.safetensors, HF
    names, separate Q/K/V) -> checkpoint.load_bert_from_local -> the HIP model reproduces the golden logits;
    split_bert_model cuts the imported model."""
    from safetensors.numpy import save_file
    from polus_amd.checkpoint import HF_LAYER_MAP, load_bert_from_local
    g, ocfg, params, hw, hb = load_case("bert_small_b3_s48")
    H = ocfg.hidden_size
    sd = {"bert.embeddings.word_embeddings.weight": params["emb.word"], "bert.embeddings.position_embeddings.weight": params["emb.pos"],
          "bert.embeddings.token_type_embeddings.weight": params["emb.type"], "bert.embeddings.LayerNorm.weight": params["emb.ln.g"],
          "bert.embeddings.LayerNorm.bias": params["emb.ln.b"], "classifier.weight": hw, "classifier.bias": hb}
    for i in range(ocfg.num_hidden_layers):
        q, o = f"bert.encoder.layer.{i}.", f"layer{i}."
        for j, nme in enumerate(("query", "key", "value")):
            sd[q + f"attention.self.{nme}.weight"] = params[o + "qkv.w"][j * H:(j + 1) * H]
            sd[q + f"attention.self.{nme}.bias"] = params[o + "qkv.b"][j * H:(j + 1) * H]
        for hf, ours in HF_LAYER_MAP:
            sd[q + hf] = params[o + ours]
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    json.dump({"vocab_size": ocfg.vocab_size, "hidden_size": H, "num_hidden_layers": ocfg.num_hidden_layers,
               "num_attention_heads": ocfg.num_attention_heads, "intermediate_size": ocfg.intermediate_size,
               "max_position_embeddings": ocfg.max_position_embeddings, "type_vocab_size": ocfg.type_vocab_size,
               "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0}, open(tmp_path / "config.json", "w"))
    m = load_bert_from_local(str(tmp_path), compute_dtype=mode, num_labels=hw.shape[0])
    logits = m(input_ids=g["ids"], attention_mask=g["mask"], token_type_ids=g["token_type"], training=False)
    assert_close(host(logits), g["logits"], 2e-4 if mode == "f32" else 3e-2, "golden logits from the imported checkpoint")
    from polus_amd.models import split_bert_model
    pre, post = split_bert_model(load_bert_from_local(str(tmp_path), compute_dtype=mode), -1)
    h = pre(input_ids=g["ids"], attention_mask=g["mask"], token_type_ids=g["token_type"]).last_hidden_state
    out = post(hidden_states=h, attention_mask=g["mask"])
    last = ob.bert_fwd(params, ocfg, g["ids"], g["mask"], g["token_type"])[0]
    assert_close(host(out.last_hidden_state), last, 2e-4 if mode == "f32" else 3e-2, "pre + post model == full model")


def test_resume_from_training_state_is_bitwise(tmp_path):
    """n1 extension: parameters + Adam moments + iteration / step / dropout counters.  3 steps, save, 2 more
    == a fresh trainer that loads the state and takes the same 2 steps (dropout on, deterministic mode)."""
    from polus_amd.checkpoint import load_training_state, save_training_state
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    g, ocfg, params, hw, hb = load_case("bert_small_b2_s16")

    def mk():
        cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                         ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size,
                         hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        m = BertModel(cfg, compute_dtype="bf16", num_labels=hw.shape[0])
        m.load_numpy_params(params, hw, hb)
        m.deterministic = True
        return m, ClassifierTrainer(m, AdamWeightDecay(learning_rate=warmup_scheduler(10, 1e-3), weight_decay_rate=0.01),
                                    SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))

    def step(t, s):
        ids, mask, tt, labels = synth_batch(ocfg, 2, 16, 4, 300 + s)
        t.step_counter += 1
        return float(t.train_step({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))
    m1, t1 = mk()
    for s in range(3):
        step(t1, s)
    p = save_training_state(t1, str(tmp_path / "run"))
    tail1 = [step(t1, s) for s in (3, 4)]
    m2, t2 = mk()
    load_training_state(t2, p)
    assert t2.optimizer.iterations == 3 and t2.step_counter == 3 and m2.dropout_step == 3
    tail2 = [step(t2, s) for s in (3, 4)]
    assert tail1 == tail2, (tail1, tail2)
    assert torch.equal(m1.arena.params, m2.arena.params)


def test_prefetch_stages_batches_in_hbm_and_confusion_matrix_on_gpu():
    """n3 / n4: Dataset.prefetch hands over device tensors (pinned staging, side-stream copies) in order;
    IConfusionMatrix accumulates device predictions exactly as the host oracle does."""
    from polus_amd.data import DataLoader
    from polus_amd.metrics import MacroF1Score
    N, C = 1000, 5
    r = np.random.Generator(np.random.PCG64(8))
    xs = r.standard_normal((N, 8)).astype(np.float32)
    ys = r.integers(0, C, size=N).astype(np.int64)

    def gen():
        for i in range(N):
            yield {"x": xs[i], "y": ys[i], "id": i}
    ds = DataLoader(gen).to_tfDataset().map(lambda d: (d["x"], d["y"], d["id"])).batch(64).prefetch(3)
    seen, yt, yp = 0, [], []
    metric = MacroF1Score(num_classes=C)
    pred_all = r.integers(0, C, size=N).astype(np.int32)
    for x, y, i in ds:
        assert x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.int32
        assert torch.equal(i.cpu(), torch.arange(seen, seen + x.shape[0], dtype=i.dtype))
        assert torch.equal(x.cpu(), torch.from_numpy(xs[seen:seen + x.shape[0]]))
        p = torch.from_numpy(pred_all[seen:seen + x.shape[0]]).cuda()
        metric.samples_from_batch((p, y))                      # the (pred, true) order ValidationDataCallback passes (polus/callbacks.py:233)
        seen += x.shape[0]
    assert seen == N
    f1 = metric.evaluate()
    assert abs(f1 - oo.macro_f1(oo.confusion_matrix(pred_all, ys, C))) < 1e-12


def test_confusion_matrix_rejects_out_of_range_on_both_paths():
    """A padding label (-100) or a wrong num_classes: tf.math.confusion_matrix raises; so do the host path
    (at once) and the device path (when the counts are read) -- the same data never gives two different metrics."""
    from polus_amd.metrics import Accuracy
    y = np.array([0, 1, 2, -100, 1, 3], np.int32)
    p = np.array([0, 1, 1, 2, 1, 0], np.int32)
    host_metric = Accuracy(num_classes=3)
    with pytest.raises(ValueError, match="outside"):
        host_metric.samples_from_batch((p, y))
    dev_metric = Accuracy(num_classes=3)
    dev_metric.samples_from_batch((torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda()))
    with pytest.raises(ValueError, match="2 label"):
        dev_metric.evaluate()
    ok = Accuracy(num_classes=4)
    ok.samples_from_batch((torch.from_numpy(p[[0, 1, 2, 4, 5]]).cuda(), torch.from_numpy(y[[0, 1, 2, 4, 5]]).cuda()))
    assert abs(ok.evaluate() - 3 / 5) < 1e-12


def test_prefetched_training_equals_unprefetched_without_host_syncs():
    """Batches staged by Dataset.prefetch are allocated under the prefetch stream; the consumer marks them as used by the
    compute stream (record_stream), so the allocator cannot hand a batch's block to the next copy while queued kernels
    (the embedding backward reads the ids late in the step) have not read it.  Several steps with no host sync in
    between must equal the same steps fed from host arrays."""
    from polus_amd.data import Dataset
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    g, ocfg, params, hw, hb = load_case("bert_small_b3_s48")
    batches = []
    for s in range(12):
        ids, mask, tt, labels = synth_batch(ocfg, 3, 48, 4, 300 + s)
        batches.append(({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))

    def run(prefetch):
        cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                         ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size)
        m = BertModel(cfg, compute_dtype="bf16", num_labels=hw.shape[0])
        m.load_numpy_params(params, hw, hb)
        m.deterministic = True
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=1e-3, weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        src = Dataset(lambda: iter(batches), len(batches))
        it = src.prefetch(2) if prefetch else src
        losses = [t.train_step(x, y) for x, y in it]          # no float(): nothing waits for the GPU between steps
        out = [float(l) for l in losses]
        torch.cuda.synchronize()
        return out, m.arena.params.clone()
    a, pa = run(False)
    b, pb = run(True)
    assert a == b and torch.equal(pa, pb)


def test_global_clipnorm_is_one_norm_over_the_applied_variables():
    """tf.clip_by_global_norm semantics (polus/training.py:187-191 post_process_grads / Keras global_clipnorm):
    one norm over every variable handed to apply_gradients -- across two arenas -- and nothing else: a frozen
    variable's (huge, stale) gradient window must not count."""
    from polus_amd.layers import Dense
    from polus_amd.models import Sequential
    from polus_amd.optimizers import Adam
    a = Sequential([Dense(4)], compute_dtype="f32", input_dim=8, name="a")
    b = Sequential([Dense(3)], compute_dtype="f32", input_dim=4, name="b")
    r = np.random.Generator(np.random.PCG64(31))
    vs = a.trainable_weights + b.trainable_weights
    p = {f"{i}": v.numpy().astype(np.float64) for i, v in enumerate(vs)}
    g = {k: r.standard_normal(w.shape) for k, w in p.items()}
    for (k, gv), v in zip(g.items(), vs):
        v.grad.copy_(dev(gv, torch.float32))
    frozen = 3                                                   # b's bias: not passed to apply_gradients
    vs[frozen].grad.fill_(1e6)
    used = [i for i in range(len(vs)) if i != frozen]
    opt = Adam(1e-2, global_clipnorm=0.5)
    opt.apply_gradients([(vs[i].grad, vs[i]) for i in used])
    gu = {str(i): g[str(i)] for i in used}
    clipped, gn = oo.clip_by_global_norm(gu, 0.5)
    assert gn > 0.5
    ref = {k: p[k].copy() for k in gu}
    oo.Adam(lr=1e-2, eps=1e-7).step(ref, clipped)
    for i in used:
        assert_close(host(vs[i].value), ref[str(i)], 2e-5, f"variable {i}")
    assert np.array_equal(vs[frozen].numpy().astype(np.float64), p[str(frozen)])


def test_loss_scalars_are_independent_values():
    """A loss kept from step k still reads step k's value after step k+1 (the reference returns tf tensors)."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    loss_fn = SparseCategoricalCrossentropy()
    r = np.random.Generator(np.random.PCG64(2))
    labels = r.integers(0, 5, size=(7,)).astype(np.int32)
    kept, ref = [], []
    for s in range(4):
        logits = r.standard_normal((7, 5)).astype(np.float32) * (s + 1)
        kept.append(loss_fn(labels, dev(logits)))
        ref.append(ol.sparse_softmax_xent_fwd(logits.astype(np.float64), labels)[0])
    assert np.allclose([float(k) for k in kept], ref, atol=1e-5)


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_graphed_step_equals_eager_step_bitwise(mode):
    """trainer.enable_step_graph(): after the eager warm-up steps the step is replayed from a captured HIP
    graph with the dropout salt and AdamW's lr / lr_t read from device memory -- same masks, same schedule:
    losses and parameters equal the eager run bit for bit, over the capture boundary and a changing batch."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    g, ocfg, params, hw, hb = load_case("bert_small_b3_s48")
    steps = 9

    def run(graphed):
        cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                         ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size,
                         hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        m = BertModel(cfg, compute_dtype=mode, num_labels=hw.shape[0])
        m.load_numpy_params(params, hw, hb)
        m.deterministic = True
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=warmup_scheduler(steps, 1e-3), weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        if graphed:
            t.enable_step_graph(warmup=2)
        losses = []
        for s in range(steps):
            ids, mask, tt, labels = synth_batch(ocfg, 3, 48, 4, 500 + s)
            losses.append(t.train_step({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))
        out = [float(l) for l in losses]          # read late: every step kept its own scalar
        torch.cuda.synchronize()
        if graphed:
            assert t._graphed.graph is not None and m.dropout_step == steps and t.optimizer.iterations == steps
        return out, m.arena.params.clone()
    eager, p_eager = run(False)
    graphed, p_graph = run(True)
    assert eager == graphed, (eager, graphed)
    assert torch.equal(p_eager, p_graph)
    assert eager[-1] < eager[0]


def test_graphed_step_survives_other_shapes_and_inference_between_replays():
    """The captured graph holds raw pointers to the shape-keyed buffers of the model, the loss and the workspace.  A
    short batch (runs eagerly), a LONGER batch (grows every cache and the workspace) and an inference call between
    replays replace those cache entries; GraphedStep keeps the captured buffers alive, so the replays that follow
    must still equal the eager run bit for bit -- with unrelated allocations in between that would land in any block
    the graph had wrongly released."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    g, ocfg, params, hw, hb = load_case("bert_small_b3_s48")
    plan = [(3, 48)] * 4 + [(2, 32), (3, 48), (5, 64), (3, 48), "infer", (3, 48), (3, 48)]

    def run(graphed):
        cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                         ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size,
                         hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        m = BertModel(cfg, compute_dtype="bf16", num_labels=hw.shape[0])
        m.load_numpy_params(params, hw, hb)
        m.deterministic = True
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=1e-3, weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        if graphed:
            t.enable_step_graph(warmup=2)
        losses, junk = [], []
        for s, item in enumerate(plan):
            if item == "infer":
                ids, mask, tt, _ = synth_batch(ocfg, 4, 40, 4, 900)
                m({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, training=False)
                continue
            B, S = item
            ids, mask, tt, labels = synth_batch(ocfg, B, S, 4, 700 + s)
            losses.append(t.train_step({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))
            # allocations that would reuse freed blocks of the captured step, filled with a poison value
            junk = [torch.full((1 << 18,), float("nan"), device=m.arena.device) for _ in range(8)]
        out = [float(l) for l in losses]
        torch.cuda.synchronize()
        del junk
        return out, m.arena.params.clone()
    eager, p_eager = run(False)
    graphed, p_graph = run(True)
    assert eager == graphed, (eager, graphed)
    assert torch.equal(p_eager, p_graph)


@pytest.mark.parametrize("mode,accum", [("bf16", 1), ("f32", 1), ("bf16", 2)])
def test_update_in_backward_equals_single_launch_bitwise(mode, accum, monkeypatch):
    """Single-process steps run the fused AdamW of each gradient window on a side stream as soon as backward
    reports it final (training._UpdateInBackward).  Same kernel and arithmetic as the one launch after backward
    (polus/training.py:191): losses, parameters, both Adam slots and the transposed bf16 shadows must come out
    bit-identical, with dropout on, a warm-up schedule, and under gradient accumulation (last micro-step only)."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    from tests.golden.make_golden import synth_batch
    g, ocfg, params, hw, hb = load_case("bert_small_b3_s48")
    steps = 6

    def run(in_backward):
        monkeypatch.setenv("POLUS_UPDATE_IN_BACKWARD", "1" if in_backward else "0")
        cfg = BertConfig(ocfg.vocab_size, ocfg.hidden_size, ocfg.num_hidden_layers, ocfg.num_attention_heads,
                         ocfg.intermediate_size, ocfg.max_position_embeddings, ocfg.type_vocab_size,
                         hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        m = BertModel(cfg, compute_dtype=mode, num_labels=hw.shape[0])
        m.load_numpy_params(params, hw, hb)
        m.deterministic = True
        t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=warmup_scheduler(steps, 1e-3), weight_decay_rate=0.01),
                              SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
        t.grad_accum_steps = accum
        losses = []
        for s in range(steps * accum):
            ids, mask, tt, labels = synth_batch(ocfg, 3, 48, 4, 700 + s)
            losses.append(t.train_step({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}, labels))
        out = [float(l) for l in losses]
        torch.cuda.synchronize()
        assert (t._updater() is not None) == in_backward
        assert t.optimizer.iterations == steps
        mm, vv = t.optimizer._slots(m.arena)
        extra = [m.arena.shadow_t.clone()] if getattr(m.arena, "shadow_t", None) is not None else []
        return out, [m.arena.params.clone(), mm.clone(), vv.clone()] + extra
    l_ref, s_ref = run(False)
    l_new, s_new = run(True)
    assert l_ref == l_new, (l_ref, l_new)
    for a, b in zip(s_ref, s_new):
        assert torch.equal(a, b)


def test_update_in_backward_is_chosen_by_tokens_per_step(monkeypatch):
    """With POLUS_UPDATE_IN_BACKWARD unset the in-backward optimizer update is taken from 6144 tokens per step on: below, a
    layer's update window outlasts the weight-gradient launch it hides under (measured on BASELINE configs[1]: 4096 tokens,
    5.56 -> 5.44 ms per step with the single launch after backward).  Both paths are the same arithmetic (test above)."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.training import ClassifierTrainer
    monkeypatch.delenv("POLUS_UPDATE_IN_BACKWARD", raising=False)
    cfg = BertConfig(512, 128, 1, 2, 256, 512, 2)
    m = BertModel(cfg, compute_dtype="bf16", num_labels=4)
    t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=1e-3), SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
    r = np.random.Generator(np.random.PCG64(5))
    for B, S, expect in ((8, 512, False), (16, 512, True), (32, 128, False)):
        ids = torch.from_numpy(r.integers(1, 512, size=(B, S)).astype(np.int32)).cuda()
        mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
        labels = torch.from_numpy(r.integers(0, 4, size=(B, S)).astype(np.int32)).cuda()
        loss = float(t.train_step({"input_ids": ids, "attention_mask": mask}, labels))
        assert np.isfinite(loss)
        assert (t._updater() is not None) == expect, (B, S)
