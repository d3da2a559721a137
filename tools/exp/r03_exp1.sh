# round 3, experiment 1: tests of the reworked host code, tile-order A/B (isolated warm / cold, and in the step)
cd $GRAFT_REPO_ROOT
python3 tools/pp_bench.py --ab POLUS_GEMM_ORDER=0,2,3,4,6 --rounds 5 --iters 10 > gpurun_out/r03_order_warm.txt 2>&1 || exit 1
python3 tools/pp_bench.py --ab POLUS_GEMM_ORDER=0,2,3,4,6 --rounds 3 --iters 6 --cold > gpurun_out/r03_order_cold.txt 2>&1 || exit 1
for o in 0 4 0 4; do
  POLUS_GEMM_ORDER=$o python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r03_bench_order$o.json 2>> gpurun_out/r03_bench_order.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/r03_bench_order$o.json'));print('order $o', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> gpurun_out/r03_order_step.txt
done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_tests1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03_tests1.log
tail -5 gpurun_out/r03_tests1.log; cat gpurun_out/r03_order_step.txt
