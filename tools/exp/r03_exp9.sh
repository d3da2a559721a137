cd $GRAFT_REPO_ROOT
python3 tools/pp_bench.py --ab POLUS_GEMM_ABLATE=0,128 --rounds 7 --iters 10 2>&1 | grep "ffn1\|du \|ABLATE"
python3 tools/pp_bench.py --ab POLUS_GEMM_ABLATE=0,128 --rounds 5 --iters 6 --cold 2>&1 | grep "ffn1\|du \|ABLATE"
for pv in 0 128 0 128 0 128; do
  POLUS_GEMM_ABLATE=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('full-drain $pv', d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
for pv in 0 1 0 1; do
  POLUS_GEMM_PERSIST=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('persist $pv', d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
