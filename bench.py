"""Headline benchmark: train samples/sec of BERT-base token classification (NER), seq 256,
64 samples per GPU, data-parallel over N MI355X of one node (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full ClassifierTrainer.train_step (forward, loss, backward, bucketed RCCL
gradient all-reduce when N > 1, fused AdamW) on one synthetic batch already resident in HBM.
Weak scaling: per-GPU batch fixed.  Rank 0 prints ONE JSON line.

`roofline` describes the dominant kernel (the bf16 MFMA GEMM of the forward Dense layers):
its launches are bracketed with HIP events during one extra instrumented step after the
timed region; achieved = algorithmic FLOPs of those launches / their summed device time.
`step_mfma_frac` is BASELINE.md's whole-step figure: samples/s/GPU x F_step(sample) / peak.
`cpu_baseline` times the NumPy oracle's identical step (fwd + bwd + AdamW) on the host cores
of this box, on a bounded sample (B=4): a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X dense MFMA (MI355X_MICROARCH.md)
VOCAB = 28996                                     # BioBERT-base-cased vocabulary
N_LABELS = 4                                      # PAD, O, B-Chemical, I-Chemical (polus/ner/utils.py:9-14)


def f_step_per_sample(L, S, H):
    """F_fwd = L*S*(24 H^2 + 4 S H), F_step = 3 F_fwd (encoder only)."""
    return 3.0 * L * S * (24.0 * H * H + 4.0 * S * H)


def synth_batch(B, S, seed, vocab=VOCAB, C=N_LABELS):
    """ids U{1000..V-1}, CLS first, SEP last-valid, lengths U{S/2..S}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(1000, vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(S // 2, S + 1, size=(B,))
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    ids[:, 0] = 101
    ids[np.arange(B), lens - 1] = 102
    ids = ids * mask
    labels = (rng.integers(0, C, size=(B, S)) * mask).astype(np.int32)
    return ids, mask, np.zeros_like(ids), labels


def cpu_baseline(S, L, H, A, I, seconds_budget=25.0):
    """The oracle's train step (NumPy/OpenBLAS, f32) on this box's host cores."""
    from oracle import bert as ob
    from oracle import optim as oo
    B = 4
    cfg = ob.BertConfig(VOCAB, H, L, A, I, 512, 2)
    params = ob.init_params(cfg, seed=1234, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(7))
    hw = (np.clip(rng.standard_normal((N_LABELS, H)), -2, 2) * 0.02).astype(np.float32)
    hb = np.zeros(N_LABELS, np.float32)
    allp = dict(params); allp["head.w"] = hw; allp["head.b"] = hb
    opt = oo.Adam(lr=5e-5, weight_decay=0.01, no_decay=[k for k in allp if oo.is_no_decay(k)])

    def step(seed):
        ids, mask, tt, labels = synth_batch(B, S, seed)
        loss, _, cache = ob.token_classifier_fwd(allp, cfg, allp["head.w"], allp["head.b"], ids, mask, labels, tt)
        opt.step(allp, ob.token_classifier_bwd(allp, cfg, allp["head.w"], cache))
        return loss

    step(0)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < seconds_budget * 0.5 and n < 8):
        step(1 + n)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"NumPy oracle, f32, B={B} S={S} BERT-base L={L}, {n} timed steps after 1 warm-up "
                      f"({dt:.1f} s), OpenBLAS threads = all cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--large", action="store_true", help="BERT-large instead of BERT-base")
    ap.add_argument("--dropout", type=float, default=0.1, help="hidden and attention dropout (HF BERT default 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from polus_amd import comm, ops
    from polus_amd.context import PolusContext
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer

    ctx = PolusContext()           # joins the torchrun rendezvous when WORLD_SIZE > 1
    world, rank = comm.size(), comm.rank()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if world == 1:
        torch.cuda.set_device(0)

    if args.large:
        H, A, I, L = 1024, 16, 4096, args.layers if args.layers != 12 else 24
    else:
        H, A, I, L = 768, 12, 3072, args.layers
    B, S = args.batch, args.seq
    cfg = BertConfig(vocab_size=VOCAB, hidden_size=H, num_hidden_layers=L, num_attention_heads=A,
                     intermediate_size=I, max_position_embeddings=512,
                     hidden_dropout_prob=args.dropout, attention_probs_dropout_prob=args.dropout)
    model = BertModel(cfg, compute_dtype=args.dtype, num_labels=N_LABELS, seed=1234)
    total_steps = max(1000, args.steps + args.warmup)
    opt = AdamWeightDecay(learning_rate=warmup_scheduler(total_steps, 5e-5), weight_decay_rate=0.01)
    loss_fn = SparseCategoricalCrossentropy(grad_dtype=model.compute_dtype)
    trainer = ClassifierTrainer(model, opt, loss_fn)
    if ctx.is_horovod_enabled():
        trainer.broadcast_init_vars()

    dev = model.arena.device
    batches = []
    for k in range(4):
        ids, mask, tt, labels = synth_batch(B, S, 42 + rank + 1000 * k)
        batches.append(({"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev),
                         "token_type_ids": torch.from_numpy(tt).to(dev)}, torch.from_numpy(labels).to(dev)))

    def one_step(k):
        x, y = batches[k % len(batches)]
        return trainer.train_step(x, y)

    first_loss = None
    for k in range(args.warmup):
        l = one_step(k)
        if first_loss is None:
            first_loss = float(l)
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = one_step(args.warmup + k)
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    last_loss = float(last)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- instrumented step: HIP events around every launch of the dominant kernel
    # (every rank takes the step -- it contains the gradient all-reduce -- only rank 0 records)
    roof = None
    rec = []
    # The dW GEMMs normally run beside the dX GEMMs on a side stream; events around a launch would
    # then time two kernels sharing the CUs, so this one step keeps everything on one stream (as
    # the rocprof summaries under profiles/ do with POLUS_OVERLAP_DW=0).
    if rank == 0:
        ops.GEMM_PROFILE = rec
    overlap = getattr(model, "overlap_dw", False)
    model.overlap_dw = False
    one_step(args.warmup + args.steps)
    torch.cuda.synchronize()
    model.overlap_dw = overlap
    ops.GEMM_PROFILE = None
    if rank == 0:
        fwd = [(e0.elapsed_time(e1) * 1e-3, fl) for (key, fl, e0, e1) in rec if key == "fwd"]
        if fwd:
            tsum, fsum = sum(t for t, _ in fwd), sum(f for _, f in fwd)
            ach = fsum / tsum / 1e12
            peak = PEAK_TFLOPS[args.dtype]
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,
                    "kernel": ("gemm_ring_kernel<bf16, K-contig, K-contig, epilogue mode 0-3>" if args.dtype == "bf16" else "gemm_kernel<f32, K-contig, K-contig>")
                              + " (every K-contiguous Dense GEMM of the step: forward QKV, out-proj, FFN1, FFN2 and their"
                                " input gradients dX = dY.W^T-shadow; the remaining GEMMs are the K-strided dW = dY^T.X)",
                    "launches": len(fwd), "avg_launch_us": round(tsum / len(fwd) * 1e6, 2),
                    "flops_per_launch": fsum / len(fwd)}
            tf = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")
            if args.dtype == "bf16" and os.path.exists(tf) and (B, S, L, H) == (64, 256, 12, 768):
                # HBM bytes per launch of this kernel from separate rocprofv3 --pmc passes (FETCH_SIZE x2
                # gfx950 correction, WRITE_SIZE) reduced by tools/hbm_traffic.py, see profiles/README.md
                roof["traffic"] = json.load(open(tf))["bytes_per_launch"]
            allg = [(e0.elapsed_time(e1) * 1e-3, fl) for (_, fl, e0, e1) in rec]
            roof["all_gemm_tflops"] = round(sum(f for _, f in allg) / sum(t for t, _ in allg) / 1e12, 2)
            roof["all_gemm_ms_per_step"] = round(sum(t for t, _ in allg) * 1e3, 3)

    if rank == 0:
        sps = world * B * args.steps / elapsed
        fstep = f_step_per_sample(L, S, H)
        out = {
            "metric": "train samples/sec BioBERT-base NER seq256", "value": round(sps, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BERT-{'large' if args.large else 'base'} (L={L},H={H},A={A},I={I},V={VOCAB}) "
                                   f"token classification C={N_LABELS}, seq_len={S}, {B} samples/GPU "
                                   f"(BASELINE.json configs[2] per-GPU shape), AdamW lr 5e-5 wd 0.01 warm-up 10%, "
                                   f"dropout {args.dropout} (hidden + attention), random-init weights",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "grad_allreduce": "bucketed RCCL all-reduce (f32, 64 MB buckets) overlapped with backward" if world > 1 else "none"},
            "step_mfma_frac": round(sps / world * fstep / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
            "step_tflops_per_gpu": round(sps / world * fstep / 1e12, 2),
            "loss_first": round(first_loss, 5) if first_loss is not None else None, "loss_last": round(last_loss, 5),
        }
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, L, H, A, I)
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.shutdown()


if __name__ == "__main__":
    main()
