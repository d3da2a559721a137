cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from polus_amd import ops
# bit-identity of the persistent form against the per-tile form on a multi-round launch, all four epilogue modes
T, dt = 16384, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda r, c: (torch.rand(r, c, device="cuda", generator=g) * 2 - 1).to(dt)
for name, N, K, kw in (("bias", 2304, 768, "bias"), ("gelu", 3072, 768, "gelu"), ("gelu-bwd", 3072, 768, "bwd"), ("resid", 3072, 320, "resid"), ("drop", 3072, 768, "drop")):
    a, b = rnd(T, K), rnd(N, K) * 0.05
    bias = torch.rand(N, device="cuda") - 0.5
    aux, res = rnd(T, N), rnd(T, N)
    outs = []
    for pers in (0, 2):
        ops.set_env("POLUS_GEMM_PERSIST", pers)
        c = torch.full((T, N), float("nan"), dtype=dt, device="cuda")
        ax = aux.clone()
        k = {"bias": dict(bias=bias), "gelu": dict(bias=bias, aux=ax, act="gelu", flags=ops.GEMM_ACT_FWD),
             "bwd": dict(aux=ax, act="gelu", flags=ops.GEMM_ACT_BWD), "resid": dict(resid=res),
             "drop": dict(bias=bias, resid=res, drop_p=0.1, seed=5)}[kw]
        ops.gemm(a, b, c, **k)
        torch.cuda.synchronize()
        outs.append((c, ax))
    ops.set_env("POLUS_GEMM_PERSIST")
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(name, "bit-identical" if same else "DIFFERENT", float(outs[0][0].float().abs().mean()))
    assert same
PY
[ $? -eq 0 ] || exit 1
python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2 --rounds 5 --iters 10 > gpurun_out/r03_persist_warm.txt 2>&1
python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2 --rounds 3 --iters 6 --cold > gpurun_out/r03_persist_cold.txt 2>&1
cat gpurun_out/r03_persist_warm.txt gpurun_out/r03_persist_cold.txt
for pv in 0 1 0 1; do
  POLUS_GEMM_PERSIST=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('persist $pv', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
