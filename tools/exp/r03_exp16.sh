cd $GRAFT_REPO_ROOT
for pv in 0 1 2 0 1 2; do POLUS_GEMM_PERSIST=$pv python3 bench.py --config c5 --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('c5 persist=$pv', d['value'], d['ms_per_step'])"; done
