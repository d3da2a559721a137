cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc1 -- python tools/gemm_bench.py --only ffn1 --iters 2 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2 -- python tools/gemm_bench.py --only ffn1 --iters 2 > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/pmc3 -- python tools/gemm_bench.py --only ffn1 --iters 2 > gpurun_out/pmc3.log 2>&1
ls gpurun_out/pmc1/*/ gpurun_out/pmc2/*/ gpurun_out/pmc3/*/ ; tail -3 gpurun_out/pmc3.log
