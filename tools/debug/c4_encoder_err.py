"""BERT-large encoder outputs (last hidden state, pooled) of the bf16 engine under either LayerNorm kernel against
the f32 engine on the same 8 x 512 inputs: relative error of each, and of one bf16 run against the other."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from polus_amd import ops
from test_configs_gpu import oracle_setup, synth, build
ocfg, params, _, _ = oracle_setup(large=True)
B, S = 8, 512
rel = lambda a, b: float((a - b).norm() / b.norm())
for seed in (41, 42, 7, 8, 100):
    ids, mask, _, _ = synth(B, S, seed)
    kw = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
    out = {}
    for mode, hw in (("f32", 1), ("bf16", 0), ("bf16", 1)):
        ops.set_env("POLUS_LN_HALFWAVE", hw)
        enc = build(ocfg, params, None, None, mode, num_labels=None)
        o = enc(**kw, training=False)
        out[(mode, hw)] = (o.last_hidden_state.float().clone(), o.pooler_output.float().clone())
        torch.cuda.synchronize(); del enc, o; torch.cuda.empty_cache()
    f = out[("f32", 1)]; a = out[("bf16", 0)]; b = out[("bf16", 1)]
    print(f"seed {seed}: hidden rel err hw0 {rel(a[0], f[0]):.4e} hw1 {rel(b[0], f[0]):.4e} hw0-vs-hw1 {rel(a[0], b[0]):.4e} | "
          f"pooled hw0 {rel(a[1], f[1]):.4e} hw1 {rel(b[1], f[1]):.4e} hw0-vs-hw1 {rel(a[1], b[1]):.4e}", flush=True)
ops.set_env("POLUS_LN_HALFWAVE")
