cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e15; rm -rf $O; mkdir -p $O
for rep in 1 2 3 4; do for v in "6 1" "0 1" "6 0" "0 0"; do set -- $v
  POLUS_GEMM_STAGGER_US=$1 POLUS_GEMM_DYNAMIC=$2 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('stagger $1 dynamic $2 rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done > $O/ab.txt
cat $O/ab.txt
