cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e7; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py tests/test_boundary_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
  POLUS_DW_DEFER_REDUCE=$1 POLUS_UPDATE_AFTER_LN=$2 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('defer-reduce $1 update-after-LN $2 rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done > $O/ab.txt
cat $O/ab.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 tools/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
