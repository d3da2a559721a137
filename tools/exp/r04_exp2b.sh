# round 4, experiment 2b: CU budget of the stream-K grouped dW launch inside the step (the side-stream overlap lives on the CUs it leaves)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e2b; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for cus in even 160 192 208 224 232 240 256; do
  if [ $cus = even ]; then export POLUS_DW_STREAMK=0; else export POLUS_DW_STREAMK=1 POLUS_DW_SK_CUS=$cus; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('dw cus $cus rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done > $O/ab_cus.txt
cat $O/ab_cus.txt
