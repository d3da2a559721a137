cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e19; rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_boundary_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do for v in "0 700" "auto 700" "auto 400" "auto 1000" "1 700"; do set -- $v
  if [ $1 = auto ]; then unset POLUS_UPDATE_IN_BACKWARD; else export POLUS_UPDATE_IN_BACKWARD=$1; fi
  POLUS_UPDATE_PARAMS_PER_TOKEN=$2 python3 bench.py --config c2 --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 update-in-backward $1 params/token $2 rep $rep: %.3f ms/step  %.1f samples/s' % (d['ms_per_step'], d['value']))"
done; done > $O/ab_c2.txt; cat $O/ab_c2.txt
