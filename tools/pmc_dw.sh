cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_dw1 -- python3 tools/gemm_bench.py --only ffn1 --iters 2 > gpurun_out/pmc_dw1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_dw2 -- python3 tools/gemm_bench.py --only ffn1 --iters 2 > gpurun_out/pmc_dw2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_dw1/*/*_counter_collection.csv gemm_ring > gpurun_out/pmc_dw1.txt
python3 tools/pmc_summary.py gpurun_out/pmc_dw2/*/*_counter_collection.csv gemm_ring > gpurun_out/pmc_dw2.txt
cat gpurun_out/pmc_dw1.txt gpurun_out/pmc_dw2.txt
