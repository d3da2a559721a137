cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_boundary_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "not loss_at_step100" 2>&1 | tail -3
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('deferred LN finalize', d['value'], d['ms_per_step'])"; done
