cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --config c2 --steps 4 --warmup 2 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 tools/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
head -3 $O/timeline.txt; tail -1 $O/timeline.txt
