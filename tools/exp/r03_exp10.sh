cd $GRAFT_REPO_ROOT
for c in c4 c5 c2; do for pv in 1 2 1 2; do
  POLUS_GEMM_PERSIST=$pv python3 bench.py --config $c --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c persist=$pv', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done
