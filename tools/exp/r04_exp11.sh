cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e11; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or pingpong" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/bench.json 2>$O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print(d['ms_per_step'], r['frac'], r['avg_launch_us'], r['avg_launch_us_with_event_pair'], r['event_pair_us'])"
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/stats.log 2>&1 || exit 1
python3 tools/kstats.py $O/stats/*/*kernel_stats.csv 8 > $O/summary.txt; head -12 $O/summary.txt
rm -rf $O/stats
