# A/B of one POLUS_* switch inside the training step on ONE box: bash tools/ab_bench.sh NAME "V1 V2 ..." [bench args]
# Cycles through the settings three times (box clocks drift); prints ms_per_step of every run.
cd $GRAFT_REPO_ROOT
NAME=$1; VALUES=$2; shift 2
for rep in 1 2 3; do
  for v in $VALUES; do
    export $NAME=$v
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || exit 1
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1]); print('$NAME=$v rep $rep: %.3f ms/step  %.1f samples/s  gemm %.1f us/launch' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_us']))"
  done
done
