"""configs[3] first-step loss of the f32 engine and of the bf16 engine with either LayerNorm kernel, over a few
input seeds: how far the bf16 loss sits from the f32 loss, and how much of that the kernel choice moves."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from polus_amd import ops
from test_configs_gpu import oracle_setup, synth, build
from polus_amd.ir.models import DualEncoder
from polus_amd.ir.training import ContrastiveLoss, EfficientDenseRetrievalTrainer, InBatchDotScores
from polus_amd.optimizers import Adam
ocfg, params, _, _ = oracle_setup(large=True)
B, S, E = 8, 512, 128
for sq, sd in ((41, 42), (7, 8), (100, 101)):
    qi, qm, _, _ = synth(B, S, sq); di, dm, _, _ = synth(B, S, sd)
    q = {"input_ids": torch.from_numpy(qi).cuda(), "attention_mask": torch.from_numpy(qm).cuda()}
    d = {"input_ids": torch.from_numpy(di).cuda(), "attention_mask": torch.from_numpy(dm).cuda()}
    res = {}
    for mode, hw in (("f32", 1), ("bf16", 0), ("bf16", 1)):
        ops.set_env("POLUS_LN_HALFWAVE", hw)
        enc = build(ocfg, params, None, None, mode, num_labels=None)
        model = DualEncoder(enc, projection_dim=E, compute_dtype=mode)
        tr = EfficientDenseRetrievalTrainer(model, InBatchDotScores(), optimizer=Adam(1e-3), loss=ContrastiveLoss())
        res[(mode, hw)] = float(tr.train_step(q, d)); torch.cuda.synchronize()
        del tr, model, enc; torch.cuda.empty_cache()
    f = res[("f32", 1)]
    print(f"seeds {sq},{sd}: f32 {f:.4f}  bf16/hw0 {res[('bf16',0)]:.4f} ({res[('bf16',0)]-f:+.4f})  bf16/hw1 {res[('bf16',1)]:.4f} ({res[('bf16',1)]-f:+.4f})", flush=True)
ops.set_env("POLUS_LN_HALFWAVE")
