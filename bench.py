"""Headline benchmark: train samples/sec of BERT-base token classification (NER), seq 256,
64 samples per GPU, data-parallel over N MI355X of one node (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # bare: the parent spawns the N ranks itself (before it touches torch or the GPU)
    python bench.py --config c2|c4|c5|c5m16  # the other BASELINE.json workloads, same JSON schema (default c3 = the headline)

A step = one full ClassifierTrainer.train_step (forward, loss, backward, bucketed RCCL
gradient all-reduce when N > 1, fused AdamW) on one synthetic batch already resident in HBM.
Weak scaling: per-GPU batch fixed.  Rank 0 prints ONE JSON line.

`roofline` describes the dominant kernel (the bf16 MFMA GEMM of the forward Dense layers):
its launches are bracketed with HIP events during one extra instrumented step after the
timed region; achieved = algorithmic FLOPs of those launches / their summed device time.
`step_mfma_frac` is BASELINE.md's whole-step figure: samples/s/GPU x F_step(sample) / peak.
`f32` is the same workload on the f32 engine (exact-f32 MFMA -- the reference computes in fp32), timed in
the same run; `loss_at_step100` is BASELINE.json's loss@step100 for both engines (dropout 0).
`cpu_baseline` times a torch-CPU eager fp32 restatement of the identical step (fwd + bwd + AdamW,
oracle/bert_torch.py) on the host cores of this box, on a bounded sample (B=8, 3 + 5 steps): a reported
baseline, not the target.  `roofline.traffic` (full default run, one GPU): HBM bytes per launch of that kernel family from two
counters-only `rocprofv3 --pmc` passes over a 2-step child run of this script, started before this process touches the GPU.
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


# BASELINE.json configs -> what one bench "step" is.  c3 is the headline (the metric is quoted on it).
WORKLOADS = {
    "c3": dict(label="BASELINE.json configs[2] per-GPU shape", large=False, batch=64, seq=256, accum=1, kind="ner"),
    "c2": dict(label="BASELINE.json configs[1]", large=False, batch=32, seq=128, accum=1, kind="ner"),
    # configs[4] names no batch size and SURVEY.md section 8 reads it both ways: row a8 has T = 32768 tokens per GPU and
    # micro-step (64 x 512, configs[2]'s per-GPU batch) = c5; the FLOP table of 8(d) assumes 64 samples as 4 x 16 = c5m16
    # (what rounds 1-2 quoted as c5)
    "c5": dict(label="BASELINE.json configs[4] per-GPU shape: seq 512, 4 micro-steps of 64 samples (configs[2]'s per-GPU batch), "
                     "one exchange + AdamW per 256 samples", large=False, batch=256, seq=512, accum=4, kind="ner"),
    "c5m16": dict(label="BASELINE.json configs[4] read with 16-sample micro-steps: 64 samples as 4 micro-steps of 16, one "
                        "exchange + AdamW per step", large=False, batch=64, seq=512, accum=4, kind="ner"),
    "c4": dict(label="BASELINE.json configs[3] per-GPU shape: dual encoder, both BERT-large encoders frozen (forward only, "
                     "polus/ir/training.py:69-75), projections + in-batch softmax CE trained",
               large=True, batch=64, seq=512, accum=1, kind="ir"),
}


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it: spawn the N ranks as fresh child processes (the parent has
    not imported torch nor touched the GPU), relay rank 0's stdout (the JSON line), fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:             # a failed rank leaves its peers inside a collective: end exactly those PIDs
                    q.terminate()
        time.sleep(0.2)
    return rc


def measure_hbm_traffic(extra_argv=()):
    """roofline.traffic, live: HBM bytes per launch of the K-contiguous GEMM family from the PMC counters, collected as the
    MI355X guide prescribes -- counters-only rocprofv3 passes (no trace domains), FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they
    do not fit one), both in KiB, FETCH_SIZE doubled (gfx950 reports wide coalesced reads at half size), WRITE_SIZE exact.  Each pass
    profiles a 2-step child run of THIS script; the children are started before this process has imported torch or touched the GPU,
    with python3 itself behind `--`.  Returns a dict, or a string saying why there is no number (the line then carries null)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return "rocprofv3 not found"
    sums, t_begin = {}, time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="polus_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-f32-leg", "--no-loss100", "--no-traffic"] + list(extra_argv)
        env = dict(os.environ, TMPDIR="/tmp", POLUS_BENCH_CHILD="1")
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=max(30.0, 330.0 - (time.time() - t_begin)))
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)          # exactly the process group this call started
                p.wait()
                return f"the {counter} pass did not finish in time"
            if rc != 0:
                return f"the {counter} pass exited with code {rc}"
            vals = []
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    n = r["Kernel_Name"]
                    if r["Counter_Name"] == counter and ("gemm_pp_kernel" in n or "gemm_ppp_kernel" in n or "gemm_ring_kernelIDF16bLb0ELb0E" in n):
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return f"the {counter} pass recorded no launch of the kernel family"
            sums[counter] = (sum(vals) / len(vals), len(vals))
        finally:
            shutil.rmtree(out, ignore_errors=True)
    (f_kib, nf), (w_kib, nw) = sums["FETCH_SIZE"], sums["WRITE_SIZE"]
    if nf != nw:
        return f"the two passes saw different launch counts ({nf}, {nw})"
    return {"bytes_per_launch": round((2.0 * f_kib + w_kib) * 1024.0), "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
            "launches_counted": nf, "seconds": round(time.time() - t_begin, 1),
            "method": "two counters-only `rocprofv3 --pmc` child passes of this script (2 steps each): FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, mean over the family's launches"}


def tf_reference_probe():
    """the reference's TF path can be timed only where `import tensorflow` works."""
    import importlib.util
    missing = [m for m in ("tensorflow", "tensorflow_addons", "horovod") if importlib.util.find_spec(m) is None]
    if "tensorflow" in missing:
        return "unavailable (import tensorflow: ModuleNotFoundError) -- cpu_baseline is the torch-CPU port"
    return ("tensorflow importable, but the reference path also needs " + ", ".join(missing) + " and HF TF-BERT: not timed"
            if missing else "tensorflow importable; the reference tree itself is not on this box: not timed")


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


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    shows all 256 hardware threads but schedules a one-GPU job on its share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(S, L, H, A, I, B=8, warm=3, timed=5, budget_s=90.0):
    """The same training step (forward, mean CE, backward, AdamW) as a torch-CPU eager fp32 restatement
    (oracle/bert_torch.py, pinned to the NumPy oracle by tests/test_oracle_golden.py) on this box's host
    cores: SURVEY.md §8(d) -- B=8, 3 warm-up + 5 timed steps, all cores."""
    import torch
    from oracle import bert as ob
    from oracle import bert_torch as bt
    from oracle import optim as oo
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = ob.BertConfig(VOCAB, H, L, A, I, 512, 2)
    params = ob.init_params(cfg, seed=1234, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(7))
    params["head.w"] = (np.clip(rng.standard_normal((N_LABELS, H)), -2, 2) * 0.02).astype(np.float32)
    params["head.b"] = np.zeros(N_LABELS, np.float32)
    p = bt.to_torch(params, torch.float32)
    opt = bt.Adam(lr=5e-5, weight_decay=0.01, no_decay=[k for k in params if oo.is_no_decay(k)])

    def step(seed):
        ids, mask, tt, labels = synth_batch(B, S, seed)
        return bt.train_step(p, cfg, opt, ids, mask, labels, tt)

    tw = time.perf_counter()
    done_warm = 0
    for k in range(warm):
        step(k)
        done_warm += 1
        if time.perf_counter() - tw > budget_s * 0.4:      # a slow host: keep the whole leg bounded
            break
    t0 = time.perf_counter()
    done = 0
    for k in range(timed):
        step(warm + k)
        done += 1
        if done >= 2 and time.perf_counter() - t0 > budget_s * 0.6:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * done / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"torch-CPU eager fp32 restatement of the step (oracle/bert_torch.py), B={B} S={S} "
                      f"BERT-base L={L}, {done} timed steps after {done_warm} warm-up ({dt:.1f} s), "
                      f"torch threads = {torch.get_num_threads()} (host shows {os.cpu_count()} hardware threads)"}


def build_trainer(args, dtype, dropout, total_steps):
    """The trainer of the selected workload: (model, trainer, step_fn(k) -> loss of the last micro-step)."""
    from polus_amd.losses import SparseCategoricalCrossentropy
    from polus_amd.models import BertConfig, BertModel
    from polus_amd.optimizers import Adam, AdamWeightDecay
    from polus_amd.schedulers import warmup_scheduler
    from polus_amd.training import ClassifierTrainer
    H, A, I, L = args.geom
    cfg = BertConfig(vocab_size=VOCAB, hidden_size=H, num_hidden_layers=L, num_attention_heads=A,
                     intermediate_size=I, max_position_embeddings=512,
                     hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    if args.kind == "ir":
        from polus_amd.ir.models import DualEncoder
        from polus_amd.ir.training import ContrastiveLoss, EfficientDenseRetrievalTrainer, InBatchDotScores
        enc = BertModel(cfg, compute_dtype=dtype, seed=1234)
        model = DualEncoder(enc, projection_dim=128, compute_dtype=dtype)
        trainer = EfficientDenseRetrievalTrainer(model, InBatchDotScores(), optimizer=Adam(1e-3), loss=ContrastiveLoss())
        return model, trainer
    model = BertModel(cfg, compute_dtype=dtype, num_labels=N_LABELS, seed=1234)
    opt = AdamWeightDecay(learning_rate=warmup_scheduler(total_steps, 5e-5), weight_decay_rate=0.01)
    trainer = ClassifierTrainer(model, opt, SparseCategoricalCrossentropy(grad_dtype=model.compute_dtype))
    trainer.grad_accum_steps = args.accum
    return model, trainer


def device_batches(args, rank, dev, n=4, seed0=42):
    """Synthetic batches resident in HBM before the timed region.  ner: (inputs, labels) per micro-step;
    ir: (query, positive document)."""
    import torch
    B, S = args.batch, args.seq
    out = []
    for k in range(n):
        ids, mask, tt, labels = synth_batch(B, S, seed0 + rank + 1000 * k)
        x = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev),
             "token_type_ids": torch.from_numpy(tt).to(dev)}
        if args.kind == "ir":
            di, dm, _, _ = synth_batch(B, S, seed0 + rank + 1000 * k + 500)
            q = {"input_ids": x["input_ids"], "attention_mask": x["attention_mask"]}
            out.append((q, {"input_ids": torch.from_numpy(di).to(dev), "attention_mask": torch.from_numpy(dm).to(dev)}))
        elif args.accum > 1:
            mb = B // args.accum
            y = torch.from_numpy(labels).to(dev)
            out.append([({kk: v[m:m + mb] for kk, v in x.items()}, y[m:m + mb]) for m in range(0, B, mb)])
        else:
            out.append((x, torch.from_numpy(labels).to(dev)))
    return out


def make_step(args, trainer, batches):
    def one_step(k):
        b = batches[k % len(batches)]
        if args.kind == "ner" and args.accum > 1:
            for x, y in b:                       # the exchange and the update happen on the last micro-step only
                loss = trainer.train_step(x, y)
            return loss
        return trainer.train_step(*b)
    return one_step


def run_leg(args, dtype, steps, warmup, ctx, world, rank):
    """W untimed + K timed steps of the workload on the `dtype` engine, then one instrumented
    step (HIP events around every GEMM launch).  Returns the numbers of the JSON line for that engine."""
    import torch
    from polus_amd import comm, ops
    H, A, I, L = args.geom
    B, S = args.batch, args.seq
    model, trainer = build_trainer(args, dtype, args.dropout, max(1000, steps + warmup))
    if ctx.is_horovod_enabled():
        trainer.broadcast_init_vars()
    if getattr(args, "graph", False) and world == 1 and args.kind == "ner" and args.accum == 1:
        trainer.enable_step_graph(warmup=min(3, max(1, warmup - 1)))
    batches = device_batches(args, rank, model.arena.device)
    one_step = make_step(args, trainer, batches)

    # N > 1: let the trainer measure, on this node, whether the GEMMs of an exchanging backward pass should leave CUs to
    # RCCL's channel kernels (untimed real training steps before the warm-up; every rank takes part)
    tuned = None
    if world > 1 and os.environ.get("POLUS_DP_TUNE", "1") != "0":
        prior = getattr(args, "dp_choice", None)
        if prior is not None:
            # a later leg of the same run (the f32 engine): the node has been measured once, take its choice over
            trainer.reserve_cus_in_backward = prior["reserve_cus_in_backward"]
            if "bucket_mb" in prior:
                trainer.bucket_mb = prior["bucket_mb"]
                trainer._reducers.clear()
            tuned = dict(prior, reused_from_first_leg=True)
        else:
            tick = [0]

            def tune_step():
                tick[0] += 1
                return one_step(10_000 + tick[0])
            tuned = trainer.tune_data_parallel(tune_step, steps=4)
            if tuned is not None:
                # what the exchange actually ran with (comm.init may have set the first; the library reads the second)
                tuned["NCCL_MAX_NCHANNELS"] = os.environ.get("NCCL_MAX_NCHANNELS")
                tuned["POLUS_GEMM_RESERVE_CUS"] = os.environ.get("POLUS_GEMM_RESERVE_CUS")
                args.dp_choice = {k: tuned[k] for k in ("reserve_cus_in_backward", "bucket_mb", "NCCL_MAX_NCHANNELS", "POLUS_GEMM_RESERVE_CUS") if k in tuned}
    first_loss = None
    for k in range(warmup):
        l = one_step(k)
        if first_loss is None:
            first_loss = float(l)
    trainer.measure_exposed = world > 1
    exposed = []
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        last = one_step(warmup + k)
        if world > 1 and getattr(trainer, "exposed_events", None) is not None:
            exposed.append(trainer.exposed_events)
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    last_loss = float(last)
    elapsed = comm.max_over_ranks(elapsed)
    trainer.measure_exposed = False

    # ---- instrumented step: HIP events (recorded on the launch stream) around every GEMM launch.
    # Every rank takes the step -- it contains the gradient all-reduce -- only rank 0 records.  The dW
    # GEMMs normally run beside the dX GEMMs on a side stream; an event pair would then time two kernels
    # sharing the CUs, and the optimizer update normally starts inside backward on a third stream; this one
    # step keeps everything on one stream (as the rocprof summaries under profiles/ do with POLUS_OVERLAP_DW=0
    # POLUS_UPDATE_IN_BACKWARD=0).
    roof, rec = None, []
    if rank == 0:
        ops.GEMM_PROFILE = rec
    enc = getattr(model, "query_encoder", model)
    overlap = getattr(enc, "overlap_dw", False)
    enc.overlap_dw = False
    trainer.update_in_backward = False
    one_step(warmup + steps)
    torch.cuda.synchronize()
    enc.overlap_dw = overlap
    trainer.update_in_backward = True
    ops.GEMM_PROFILE = None
    peak = PEAK_TFLOPS[dtype]
    if rank == 0:
        fwd = [(e0.elapsed_time(e1) * 1e-3, fl) for (key, fl, e0, e1) in rec if key == "fwd"]
        if fwd:
            tsum, fsum = sum(t for t, _ in fwd), sum(f for _, f in fwd)
            ach = fsum / tsum / 1e12
            kern = ("gemm_pp_kernel<256 x 256 | 256 x 192, epilogue mode 0-3> (ping-pong; the ring kernels gemm_ring_kernel<bf16> / "
                    "gemm_ring128_kernel where a launch would leave CUs idle)" if dtype == "bf16" else "gemm_kernel<f32, K-contig, K-contig> (exact-f32 MFMA 16x16x4)")
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,
                    "kernel": kern + " -- every K-contiguous Dense GEMM of the step: forward QKV, out-proj, FFN1, FFN2 and "
                              "their input gradients dX = dY.W^T-shadow; the remaining GEMMs are the K-strided dW = dY^T.X",
                    "launches": len(fwd), "avg_launch_us": round(tsum / len(fwd) * 1e6, 2),
                    "flops_per_launch": fsum / len(fwd)}
            live = getattr(args, "traffic", None) if dtype == "bf16" else None
            if isinstance(live, dict):
                roof["traffic"] = live["bytes_per_launch"]
                roof["traffic_measurement"] = live
            elif isinstance(live, str):
                roof["traffic_note"] = "no live number: " + live
            tf = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
            if dtype == "bf16" and os.path.exists(tf) and args.config == "c3" and (B, S, L, H) == (64, 256, 12, 768):
                # NOT measured in this run: HBM bytes per launch of this kernel family from separate
                # rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), tools/hbm_traffic.py
                roof["traffic_stored_profile"] = {"bytes_per_launch": json.load(open(tf))["bytes_per_launch"],
                                                  "source": f"profiles/{TRAFFIC_PROFILE} (separate rocprofv3 --pmc passes; not measured in this run)"}
            allg = [(e0.elapsed_time(e1) * 1e-3, fl) for (_, fl, e0, e1) in rec]
            roof["all_gemm_tflops"] = round(sum(f for _, f in allg) / sum(t for t, _ in allg) / 1e12, 2)
            roof["all_gemm_ms_per_step"] = round(sum(t for t, _ in allg) * 1e3, 3)
    sps = world * B * steps / elapsed
    fstep = f_step_per_sample(L, S, H) * (2.0 / 3.0 if args.kind == "ir" else 1.0)   # ir: two encoder FORWARD passes per pair
    out = {"value": round(sps, 2), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "step_mfma_frac": round(sps / world * fstep / (peak * 1e12), 4), "mfma_peak_tflops": peak,
           "step_tflops_per_gpu": round(sps / world * fstep / 1e12, 2),
           "loss_first": round(first_loss, 5) if first_loss is not None else None, "loss_last": round(last_loss, 5)}
    if roof:
        out["roofline"] = roof
    if tuned is not None:
        out["dp_tuning"] = tuned
    if world > 1:
        # what the exchange costs: (a) the tail behind backward that nothing overlapped (HIP events, mean over the timed
        # steps, max over ranks); (b) the same steps with every rank training alone -- no exchange at all (the replicas
        # diverge from here on: this is the last thing this leg's trainer does)
        ex = [e0.elapsed_time(e1) for e0, e1 in exposed]
        out["exposed_comm_ms"] = round(comm.max_over_ranks(sum(ex) / max(len(ex), 1)), 3)
        trainer.use_horovod = False
        n_nc = max(3, min(steps, 10))
        one_step(0)
        torch.cuda.synchronize()
        comm.barrier()
        t0 = time.perf_counter()
        for k in range(n_nc):
            one_step(1 + k)
        torch.cuda.synchronize()
        out["ms_per_step_nocomm"] = round(comm.max_over_ranks(time.perf_counter() - t0) / n_nc * 1e3, 3)
        comm.barrier()
    del trainer, model, batches
    torch.cuda.empty_cache()
    return out


LOSS100_FIXTURE = os.path.join(ROOT, "tests", "golden", "loss100_bert_base_b64_s256.npz")
TRAFFIC_PROFILE = "r04_gemm_hbm_traffic.json"


def loss_at_step100(args, rank):
    """BASELINE.json's "loss@step100": 100 AdamW steps (warm-up 10 %, lr 5e-5, wd 0.01) of the headline shape
    with dropout 0 -- TF's dropout stream cannot be matched, so parity runs have none -- from the same
    initial weights over the same 8 recurring synthetic batches, on the f32 engine (exact-f32 MFMA) and on the bf16
    engine, next to the ORACLE's curve for the identical run (float64 torch-CPU restatement, committed fixture
    tests/golden/loss100_bert_base_b64_s256.npz made by tests/golden/make_loss100.py; the fixture is data, nothing
    of oracle/ runs here)."""
    import torch
    out = {}
    for dtype in ("f32", "bf16"):
        model, trainer = build_trainer(args, dtype, 0.0, 100)
        model.deterministic = True
        batches = device_batches(args, rank, model.arena.device, n=8, seed0=100)
        curve = [float(trainer.train_step(*batches[s % 8])) for s in range(100)]
        out[dtype] = round(curve[-1], 5)
        out[dtype + "_step1"] = round(curve[0], 5)
        del trainer, model, batches
        torch.cuda.empty_cache()
    out["abs_diff"] = round(abs(out["bf16"] - out["f32"]), 5)
    if os.path.exists(LOSS100_FIXTURE):
        z = np.load(LOSS100_FIXTURE, allow_pickle=False)
        ref = z["loss"]
        if len(ref) >= 100:
            out["oracle"] = round(float(ref[99]), 5)
            out["oracle_step1"] = round(float(ref[0]), 5)
            out["f32_minus_oracle"] = round(out["f32"] - float(ref[99]), 5)
            out["bf16_minus_oracle"] = round(out["bf16"] - float(ref[99]), 5)
            out["oracle_source"] = "tests/golden/loss100_bert_base_b64_s256.npz (oracle/bert_torch.py in float64, tests/golden/make_loss100.py)"
    out["config"] = "dropout 0, deterministic reductions, same weights and batches for both engines and for the oracle"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(WORKLOADS), help="BASELINE.json workload (c3 = the headline)")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the workload's)")
    ap.add_argument("--seq", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--large", action="store_true", help="BERT-large instead of BERT-base")
    ap.add_argument("--dropout", type=float, default=0.1, help="hidden and attention dropout (HF BERT default 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the reference-precision (f32 engine) leg")
    ap.add_argument("--no-loss100", action="store_true", help="skip the 2 x 100-step loss@step100 runs")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (launch-bound shapes, one GPU)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: nothing has touched torch or the GPU in this process
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    args.traffic = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_traffic and "POLUS_BENCH_CHILD" not in os.environ and
            args.config == "c3" and args.dtype == "bf16" and args.batch is None and args.seq is None and args.layers == 12 and not args.large and
            not (args.no_cpu_baseline or args.no_f32_leg or args.no_loss100) and "rocprof" not in os.environ.get("LD_PRELOAD", "")):
        # (only the full default run -- what the driver executes -- pays for the two passes; the A/B and profile scripts shorten the run)
        args.traffic = measure_hbm_traffic()          # before torch is imported: the children are the first to touch the GPU

    import torch
    from polus_amd import comm
    from polus_amd.context import PolusContext

    ctx = PolusContext()           # joins the torchrun rendezvous when WORLD_SIZE > 1
    world, rank = comm.size(), comm.rank()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch bare (`python bench.py --gpus N`) or with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if world == 1:
        torch.cuda.set_device(0)

    wl = WORKLOADS[args.config]
    args.kind, args.accum = wl["kind"], wl["accum"]
    args.batch = wl["batch"] if args.batch is None else args.batch
    args.seq = wl["seq"] if args.seq is None else args.seq
    large = args.large or wl["large"]
    if large:
        args.geom = (1024, 16, 4096, args.layers if args.layers != 12 else 24)
    else:
        args.geom = (768, 12, 3072, args.layers)
    H, A, I, L = args.geom
    B, S = args.batch, args.seq
    if args.config != "c3":                       # the secondary workloads: the bf16 leg and its roofline only
        args.no_f32_leg = args.no_loss100 = args.no_cpu_baseline = True
        args.dropout = 0.0 if args.kind == "ir" else args.dropout
    plane = comm.data_plane_info() if world > 1 else None

    head = run_leg(args, args.dtype, args.steps, args.warmup, ctx, world, rank)
    # the reference computes in fp32 throughout (polus/models.py:197): the same workload on the f32 engine
    # (exact-f32 MFMA), timed in the same run on every rank; fewer steps, it is ~6x slower
    f32 = None
    if args.dtype == "bf16" and not args.no_f32_leg:
        f32 = run_leg(args, "f32", max(3, min(args.steps, 6)), 2, ctx, world, rank)
    l100 = None
    if world == 1 and not args.no_loss100 and args.config == "c3" and (B, S) == (64, 256) and not large:
        l100 = loss_at_step100(args, rank)

    if rank == 0:
        metric = {"c3": "train samples/sec BioBERT-base NER seq256", "c2": "train samples/sec BioBERT-base NER seq128 bs32",
                  "c5": "train samples/sec BioBERT-base NER seq512 grad-accum x4",
                  "c5m16": "train samples/sec BioBERT-base NER seq512 grad-accum x4 (micro-batch 16)",
                  "c4": "train query-document pairs/sec PubMedBERT-large dual encoder seq512"}[args.config]
        what = (f"dual encoder (two frozen encoder passes + projections E=128 + in-batch softmax CE, Adam 1e-3), seq_len={S}, {B} pairs/GPU"
                if args.kind == "ir" else
                f"token classification C={N_LABELS}, seq_len={S}, {B} samples/GPU" +
                (f" as {args.accum} micro-steps of {B // args.accum}" if args.accum > 1 else "") +
                f", AdamW lr 5e-5 wd 0.01 warm-up 10%, dropout {args.dropout} (hidden + attention)")
        out = {
            "metric": metric, "value": head["value"], "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BERT-{'large' if large else 'base'} (L={L},H={H},A={A},I={I},V={VOCAB}) {what} "
                                   f"({wl['label']}), random-init weights",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "grad_allreduce": (f"bucketed RCCL {'reduce-scatter (f32, 64 MB buckets) overlapped with backward -> AdamW on the owned slices -> all-gather of the parameters' if os.environ.get('POLUS_DP_MODE', 'allreduce') == 'rs' else 'all-reduce (f32, 64 MB buckets) fired inside backward, the fused AdamW of each bucket queued behind it'}; data plane: {comm._STATE['backend']}") if world > 1 else "none"},
            "step_mfma_frac": head["step_mfma_frac"], "step_tflops_per_gpu": head["step_tflops_per_gpu"],
            "loss_first": head["loss_first"], "loss_last": head["loss_last"],
            "tf_reference": tf_reference_probe(),
        }
        if world > 1:
            out.update(rccl_ranks=plane["ranks_seen"], data_plane=plane["data_plane"],
                       exposed_comm_ms=head.get("exposed_comm_ms"), ms_per_step_nocomm=head.get("ms_per_step_nocomm"),
                       dp_tuning=head.get("dp_tuning"),
                       multi_gpu_parity="unpinned: no multi-GPU box in the build environment; two-rank equivalence is tested "
                                        "over gloo (tests/test_distributed_cpu.py, tests/test_dp_gpu.py)")
            if "rccl_comm_count" in plane:
                out["rccl_comm_count"] = plane["rccl_comm_count"]
        if "roofline" in head:
            out["roofline"] = head["roofline"]
        if f32 is not None:
            out["f32"] = dict(f32, unit="samples/s", dtype="f32",
                              note="same workload on the f32 engine (exact-f32 MFMA, the reference's arithmetic type); "
                                   "step_mfma_frac and roofline.frac are against the 157.3 TFLOP/s fp32 matrix peak")
        if l100 is not None:
            out["loss_at_step100"] = l100
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, L, H, A, I)
        elif world > 1:
            # timed on rank 0 at N = 1 only (a host-core measurement beside N busy ranks would be noise); same schema key
            out["cpu_baseline"] = {"value": None, "note": "timed at N=1 only: see the cpu_baseline of the n_gpus=1 line of the same build"}
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.shutdown()


if __name__ == "__main__":
    main()
