# Wall step time on ONE box: default, both overlaps off (compare with the single-stream kernel sum), and HIP-graph replay.
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-f32-leg --no-loss100 "$@" 2> gpurun_out/gap_tmp.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step' % d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default:                         $(run)"
  echo "no side streams (dW, update):    $(POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 run)"
  echo "HIP-graph replay (--graph):      $(run --graph)"
done
