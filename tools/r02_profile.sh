# Round-2 measurement set (run on the GPU box from the repo root): bench line, per-kernel stats, HBM traffic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02_stats gpurun_out/r02_stats_ov gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
python3 bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err || exit 1
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r02_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats_ov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r02_stats_ov.log 2>&1 || exit 1
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r02_pmc_fetch.log 2>&1 || exit 1
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r02_pmc_write.log 2>&1 || exit 1
rm -rf gpurun_out/r02_pmc_mfma
POLUS_OVERLAP_DW=0 POLUS_UPDATE_IN_BACKWARD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/r02_pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-loss100 > gpurun_out/r02_pmc_mfma.log 2>&1 || exit 1
python3 tools/mfma_util.py gpurun_out/r02_pmc_mfma > gpurun_out/r02_mfma_utilisation.txt
python3 tools/kstats.py gpurun_out/r02_stats/*/*kernel_stats.csv 8 > gpurun_out/r02_step_kernel_stats_summary.txt
python3 tools/kstats.py gpurun_out/r02_stats_ov/*/*kernel_stats.csv 8 > gpurun_out/r02_step_kernel_stats_overlap_summary.txt
python3 tools/hbm_traffic.py gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write gpurun_out/r02_gemm_hbm_traffic.json
cat gpurun_out/r02_bench_line.json
