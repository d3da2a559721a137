cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
head -2 $O/trace/*/*kernel_trace.csv
python3 tools/timeline.py $O/trace > $O/timeline.txt
tail -5 $O/timeline.txt
rm -rf $O/trace
