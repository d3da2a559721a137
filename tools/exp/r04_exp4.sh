# round 4, experiment 4: grouped dW with the same number of workgroups on every XCD
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e4; rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "dense_bwd_params" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python3 tools/exp/shadow_probe.py 2>&1 | grep -v amdgpu > $O/shadow_probe.txt; cat $O/shadow_probe.txt
for rep in 1 2 3; do for v in "1 0" "0 0" "1 1:232" "1 1:240" "0 1:240"; do set -- $v; export POLUS_UPDATE_AFTER_LN=$1; sk=${2%%:*}; cus=${2##*:}
  if [ $sk = 1 ]; then export POLUS_DW_STREAMK=1 POLUS_DW_SK_CUS=$cus; else unset POLUS_DW_STREAMK POLUS_DW_SK_CUS; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('update-after-LN $1 streamk $2 rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done > $O/ab.txt
cat $O/ab.txt
unset POLUS_DW_STREAMK POLUS_DW_SK_CUS; export POLUS_UPDATE_AFTER_LN=1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 tools/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
