cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "persistent" 2>&1 | tail -2
python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1 --rounds 5 --iters 10 2>&1 | grep "ffn1\|du \|all"
python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1 --rounds 3 --iters 6 --cold 2>&1 | grep "ffn1\|du \|all"
for pv in 0 1 0 1; do
  POLUS_GEMM_PERSIST=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('persist $pv', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
