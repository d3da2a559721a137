cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e18; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "split or pingpong" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do for v in 0 1; do POLUS_GEMM_PP_SPLIT=$v python3 bench.py --config c2 --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 pp-split $v rep $rep: %.3f ms/step  %.1f samples/s' % (d['ms_per_step'], d['value']))"; done; done > $O/ab_c2.txt; cat $O/ab_c2.txt
for rep in 1 2; do for v in 0 1; do POLUS_GEMM_PP_SPLIT=$v python3 bench.py --config c5m16 --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c5m16 pp-split $v rep $rep: %.3f ms/step  %.1f samples/s' % (d['ms_per_step'], d['value']))"; done; done > $O/ab_c5m16.txt; cat $O/ab_c5m16.txt
