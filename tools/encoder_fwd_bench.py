"""Forward-only throughput of a BERT encoder on the bf16 engine (what the reference's dual-encoder trainer runs through its
frozen encoders, polus/ir/training.py:69-75): python tools/encoder_fwd_bench.py [--large] [--batch B] [--seq S]."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polus_amd.models import BertConfig, BertModel
ap = argparse.ArgumentParser()
ap.add_argument("--large", action="store_true"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--seq", type=int, default=512)
a = ap.parse_args()
H, A, I, L = (1024, 16, 4096, 24) if a.large else (768, 12, 3072, 12)
cfg = BertConfig(vocab_size=28996, hidden_size=H, num_hidden_layers=L, num_attention_heads=A, intermediate_size=I, max_position_embeddings=512)
m = BertModel(cfg, compute_dtype="bf16", seed=1)
g = torch.Generator().manual_seed(0)
ids = torch.randint(1000, 28996, (a.batch, a.seq), generator=g, dtype=torch.int32).cuda()
mask = torch.ones(a.batch, a.seq, dtype=torch.int32).cuda()
x = {"input_ids": ids, "attention_mask": mask}
for _ in range(3):
    m(**x, training=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    m(**x, training=False)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n * 1e-3
flops = a.batch * L * a.seq * (24.0 * H * H + 4.0 * a.seq * H)
print(f"BERT-{'large' if a.large else 'base'} forward B={a.batch} S={a.seq}: {t*1e3:.3f} ms  {a.batch/t:.1f} samples/s  {flops/t/1e12:.0f} TFLOP/s ({flops/t/2.5e15:.3f} of the bf16 MFMA peak)")
