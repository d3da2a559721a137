# Clean single-stream kernel durations of the LayerNorm-backward reduce shapes (run on the GPU box):
# blocks 1024 / two stages (round 1), 512 / two stages, 512 / one stage.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1024 256" "512 256" "512 512"; do
  set -- $cfg
  export POLUS_LN_BWD_BLOCKS=$1 POLUS_LN_FIN_SINGLE=$2
  bash tools/quick_stats.sh lnfin_$1_$2 > /dev/null 2>&1 || exit 1
  echo "== POLUS_LN_BWD_BLOCKS=$1 POLUS_LN_FIN_SINGLE=$2"
  grep "ln_bwd_hw\|colsum_finalize\|total kernel" gpurun_out/lnfin_$1_$2_summary.txt
done
