cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python3 bench.py --config c4 --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('c4', d['value'], d['ms_per_step'], d['step_mfma_frac'])"; done
