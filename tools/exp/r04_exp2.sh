# round 4, experiment 2: stream-K remainder for the grouped dW launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e2; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "dense_bwd_params_grouped" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python3 tools/dw_bench.py 2>&1 | grep -v amdgpu > $O/dw_bench.txt
cat $O/dw_bench.txt
bash tools/ab_bench.sh POLUS_DW_STREAMK "0 1" > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
