cd $GRAFT_REPO_ROOT
for pv in 1 2 1 2 1 2; do
  POLUS_GEMM_PERSIST=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('c3 persist=$pv', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
