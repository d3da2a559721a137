cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e17; rm -rf $O; mkdir -p $O
POLUS_PP_LATE=1 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "pingpong" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
{ python3 tools/pp_bench.py --ab POLUS_PP_LATE=0,1 --rounds 4 --iters 10 2>&1 | grep -v amdgpu; python3 tools/pp_bench.py --cold --ab POLUS_PP_LATE=0,1 --rounds 3 --iters 8 2>&1 | grep -v amdgpu; } > $O/pp.txt; cat $O/pp.txt
bash tools/ab_bench.sh POLUS_PP_LATE "0 1" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
