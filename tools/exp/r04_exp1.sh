# round 4, experiment 1: cache policy of the GELU-epilogue launches (POLUS_EXP bits: 1 = dU reads its pre-activations
# non-temporally, 2 = FFN1's pre-activation store plain instead of non-temporal, 4 / 8 = FFN1 / dU C store non-temporal)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e1; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or pingpong" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
{ echo "# tools/pp_bench.py --cold --only ffn1,du --ab POLUS_EXP=0,1,2,4,8,9"; python3 tools/pp_bench.py --cold --only ffn1,du --ab POLUS_EXP=0,1,2,4,8,9 --rounds 4 --iters 8 2>&1 | grep -v amdgpu; } > $O/pp_cold.txt
cat $O/pp_cold.txt
bash tools/ab_bench.sh POLUS_EXP "0 1 4 8 9" > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
