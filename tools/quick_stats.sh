# per-kernel stats of a short bench run on ONE stream (no side-stream dW, no in-backward optimizer update) (run on the GPU box): bash tools/quick_stats.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-quick}; shift
rm -rf gpurun_out/${TAG}_stats
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 "$@" > gpurun_out/${TAG}_stats.log 2>&1 || exit 1
python3 tools/kstats.py gpurun_out/${TAG}_stats/*/*kernel_stats.csv 8 > gpurun_out/${TAG}_summary.txt
cat gpurun_out/${TAG}_summary.txt
