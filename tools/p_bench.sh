for m in 0 2; do echo "POLUS_GEMM_P=$m"; POLUS_GEMM_P=$m timeout -k 10 200 python tools/gemm_bench.py 2>&1 | cut -c1-64 | grep -v "dW\|amdgpu.ids"; done
