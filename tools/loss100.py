"""BASELINE.json: "loss@step100".  100 AdamW steps of the headline shape (BERT-base, 64 x 256 tokens,
dropout 0, same initial weights and the same 8 recurring synthetic batches) on the f32 engine (the one
pinned to the oracle by tests/test_model_gpu.py::test_loss_trajectory_100_steps) and on the bf16 engine."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fullsize_gpu import _batch, _model
from polus_amd.losses import SparseCategoricalCrossentropy
from polus_amd.optimizers import AdamWeightDecay
from polus_amd.schedulers import warmup_scheduler
from polus_amd.training import ClassifierTrainer

curves = {}
for dt in ("f32", "bf16"):
    m = _model(dt)
    m.deterministic = True
    t = ClassifierTrainer(m, AdamWeightDecay(learning_rate=warmup_scheduler(100, 5e-5), weight_decay_rate=0.01),
                          SparseCategoricalCrossentropy(grad_dtype=m.compute_dtype))
    batches = [_batch(100 + k) for k in range(8)]
    curves[dt] = [float(t.train_step(*batches[s % 8])) for s in range(100)]
    del m, t
    torch.cuda.empty_cache()
a, b = np.array(curves["f32"]), np.array(curves["bf16"])
for s in (0, 9, 24, 49, 74, 99):
    print(f"step {s + 1:3d}: f32 {a[s]:.5f}   bf16 {b[s]:.5f}   diff {b[s] - a[s]:+.5f}")
print(f"max |bf16 - f32| over 100 steps: {np.abs(a - b).max():.5f}; at step 100: {abs(a[-1] - b[-1]):.5f}")
