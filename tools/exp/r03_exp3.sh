cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_tests2.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_tests2.log
tail -12 gpurun_out/r03_tests2.log
for c in c2 c5 c4; do
  python3 bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r03_bench_$c.json 2> gpurun_out/r03_bench_$c.err || { tail -5 gpurun_out/r03_bench_$c.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/r03_bench_$c.json'));print('$c', d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['frac'], d['roofline']['launches'])"
done
