cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e9; rm -rf $O; mkdir -p $O
bash tools/ab_bench.sh POLUS_DW_AFTER_DX "0 1" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 tools/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
