cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
python3 bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --config c2 --steps 5 --warmup 2 > $O/stats_c2.log 2>&1 || exit 1
python3 tools/kstats.py $O/stats_c2/*/*kernel_stats.csv 8 > $O/c2_kernel_stats_summary.txt
rm -rf $O/stats_c2
python3 -c "import json;d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]);print('c2', d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['frac'])"
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
