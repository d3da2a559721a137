cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e16; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or layer_norm or ln" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python3 tools/ln_bench.py 2>&1 | grep -v amdgpu | tail -7
bash tools/ab_bench.sh POLUS_LN_FWD_WAVES "16 4" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
