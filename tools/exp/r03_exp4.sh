cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > gpurun_out/r03_attn_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_attn_tests.log
tail -25 gpurun_out/r03_attn_tests.log
grep -q "rc=0" gpurun_out/r03_attn_tests.log || exit 1
python3 tools/attn_bench.py > gpurun_out/r03_attn_bench.txt 2>&1 && python3 tools/attn_bench.py --seq 512 --batch 16 >> gpurun_out/r03_attn_bench.txt 2>&1
cat gpurun_out/r03_attn_bench.txt
