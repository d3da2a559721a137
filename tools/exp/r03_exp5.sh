cd $GRAFT_REPO_ROOT
for c in c2 c5 c3; do for ov in 1 0 1 0; do
  POLUS_OVERLAP_DW=$ov python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c overlap_dw=$ov', d['value'], d['ms_per_step'])"
done; done
