cd $GRAFT_REPO_ROOT
for c in c3 c2; do for v in 0 1 0 1 0 1; do
  POLUS_LN_DEFER_FINALIZE=$v python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c POLUS_LN_DEFER_FINALIZE=$v', d['value'], d['ms_per_step'])"
done; done
