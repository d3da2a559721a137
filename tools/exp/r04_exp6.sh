cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e6; rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or pingpong" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
{ python3 tools/pp_bench.py --cold --only ffn1,du --ab POLUS_GEMM_DYNAMIC=0,1 --rounds 4 --iters 8 2>&1 | grep -v amdgpu; } > $O/pp_cold.txt; cat $O/pp_cold.txt
bash tools/ab_bench.sh POLUS_GEMM_DYNAMIC "0 1" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 tools/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
