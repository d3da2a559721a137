"""Generates tests/golden/loss100_bert_base_b64_s256.npz: BASELINE.json's "loss@step100" at the headline
shape from the ORACLE (oracle/bert_torch.py, the torch-CPU restatement pinned to the NumPy oracle and through it
to the HF PyTorch twin), in float64.

    python tests/golden/make_loss100.py [--threads 4] [--steps 100]

What is trained (identical to bench.py::loss_at_step100, which runs it on the HIP engines):
  * BERT-base (L=12, H=768, A=12, I=3072, V=28996) + Dense(768 -> 4) token head, dropout 0;
  * initial weights: encoder = N(0, 0.02) truncated at 2 sigma from PCG64(1234) in arena order (the recipe of
    polus_amd.models.BertModel(seed=1234) and of oracle.bert.init_params -- the GPU test asserts the engine's
    initial weights equal these bit for bit); head = Glorot-uniform from PCG64(crc32("head:4:768")), bias 0
    (polus_amd.layers.Dense.build);
  * 8 recurring synthetic batches of 64 x 256: bench.synth_batch(64, 256, 100 + 1000 k), k = step mod 8;
  * AdamWeightDecay, lr = warmup_scheduler(100, 5e-5) (10 % linear warm-up, linear decay to 1e-7),
    wd 0.01 on matrices, beta (0.9, 0.999), eps 1e-7 (Keras placement), mean sparse CE over all positions.

The 64-sample batch is evaluated as 4 micro-batches of 16 whose gradients are averaged: the loss is a mean over
equally many positions per micro-batch, so this IS the whole-batch gradient (float64: the regrouping of the sum
is below 1e-15); it keeps the autograd graph at ~10 GB instead of ~40 GB.

Takes ~3 h on 4 threads.  Writes the loss of every step (float64) and a checksum of the final weights.
"""
import argparse
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, S, MICRO, STEPS, LR = 64, 256, 16, 100, 5e-5
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss100_bert_base_b64_s256.npz")


def initial_weights():
    """The float32 initial weights of the run (see the module docstring), as an oracle-named dict."""
    from bench import N_LABELS, VOCAB
    from oracle import bert as ob
    cfg = ob.BertConfig(VOCAB, 768, 12, 12, 3072, 512, 2)
    params = ob.init_params(cfg, seed=1234, dtype=np.float32)
    H = cfg.hidden_size
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(f"head:{N_LABELS}:{H}".encode())))
    lim = np.sqrt(6.0 / (H + N_LABELS))
    params["head.w"] = rng.uniform(-lim, lim, size=(N_LABELS, H)).astype(np.float32)
    params["head.b"] = np.zeros(N_LABELS, np.float32)
    return cfg, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--steps", type=int, default=STEPS)
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    import torch
    from bench import synth_batch
    from oracle import bert_torch as bt
    from oracle import optim as oo
    torch.set_num_threads(args.threads)
    cfg, params = initial_weights()
    p = bt.to_torch(params, torch.float64)
    opt = bt.Adam(lr=lambda t: oo.warmup_linear_lr(t, STEPS, LR), weight_decay=0.01,
                  no_decay=[k for k in params if oo.is_no_decay(k)])
    batches = [synth_batch(B, S, 100 + 1000 * k) for k in range(8)]
    curve = []
    t0 = time.time()
    for s in range(args.steps):
        ids, mask, tt, labels = batches[s % 8]
        loss = 0.0
        for m in range(0, B, MICRO):
            sl = slice(m, m + MICRO)
            l, _ = bt.token_classifier_loss(p, cfg, ids[sl], mask[sl], labels[sl], tt[sl])
            (l * (MICRO / B)).backward()          # gradients accumulate in .grad: the whole-batch mean
            loss += float(l.detach()) * (MICRO / B)
        opt.step(p)
        curve.append(loss)
        print(f"step {s + 1:3d}  loss {loss:.9f}  ({time.time() - t0:.0f} s)", flush=True)
        np.savez(args.out + ".partial.npz", loss=np.asarray(curve, np.float64))
    final_sq = {k: float((v.detach() ** 2).sum()) for k, v in p.items()}
    np.savez(args.out, loss=np.asarray(curve, np.float64), batch=B, seq=S, steps=len(curve), lr=LR,
             dtype="float64", micro_batch=MICRO,
             final_weight_sqnorm=np.asarray([final_sq[k] for k in sorted(final_sq)], np.float64),
             final_weight_names=np.asarray(sorted(final_sq)))
    os.remove(args.out + ".partial.npz")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
