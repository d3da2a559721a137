# configs[4] in its two readings (bench.py WORKLOADS c5 / c5m16): bench line + per-kernel stats.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for c in c5 c5m16; do
  python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 bench.py --config $c --steps 5 --warmup 2 > $O/stats_$c.log 2>&1 || exit 1
  python3 tools/kstats.py $O/stats_$c/*/*kernel_stats.csv 8 > $O/${c}_kernel_stats_summary.txt
  rm -rf $O/stats_$c
done
cat $O/bench_c5.json
