# Round-4 measurement set (run on the GPU box from the repo root): bench line, per-kernel stats, HBM traffic (tile order 0 vs 4),
# MFMA utilisation + wave-cycle split, attention counters, the secondary workloads.  Every rocprofv3 line puts python3 right
# behind `--` (no env / shell hop: the profiler's preloaded library has already initialised the GPU).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
rm -rf $O && mkdir -p $O
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100"
P="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-loss100"
python3 bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
export POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $P > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $P > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_mfma -- python3 $P > $O/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_attn -- python3 tools/attn_bench.py > $O/pmc_attn.log 2>&1 || exit 1
unset POLUS_OVERLAP_DW POLUS_UPDATE_IN_BACKWARD
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ov -- python3 $B > $O/stats_ov.log 2>&1 || exit 1
python3 tools/timeline.py $O/stats_ov > $O/step_timeline_overlap.txt
python3 tools/mfma_util.py $O/pmc_mfma > $O/mfma_utilisation.txt
python3 tools/kstats.py $O/stats/*/*kernel_stats.csv 8 > $O/step_kernel_stats_summary.txt
python3 tools/kstats.py $O/stats_ov/*/*kernel_stats.csv 8 > $O/step_kernel_stats_overlap_summary.txt
python3 tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/gemm_hbm_traffic.json
for f in $O/pmc_attn/*/*counter_collection.csv; do python3 tools/pmc_summary.py $f attn > $O/attention_pmc.txt; done
for c in c2 c5 c5m16 c4; do
  python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 bench.py --config $c --steps 5 --warmup 2 > $O/stats_$c.log 2>&1 || exit 1
  python3 tools/kstats.py $O/stats_$c/*/*kernel_stats.csv 8 > $O/${c}_kernel_stats_summary.txt
done
cp $O/stats/*/*kernel_stats.csv $O/step_kernel_stats.csv
rm -rf $O/stats $O/stats_ov $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_attn $O/stats_c2 $O/stats_c4 $O/stats_c5 $O/stats_c5m16
cat $O/bench_line.json; cat $O/step_kernel_stats_summary.txt
