cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2prof; rm -rf $O; mkdir -p $O
export POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/stats.log 2>&1 || exit 1
python3 tools/kstats.py $O/stats/*/*kernel_stats.csv 8 > $O/c2_serial_summary.txt
rm -rf $O/stats
cat $O/c2_serial_summary.txt | cut -c1-130
python3 bench.py --config c2 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | cut -c1-160
