for a in 0 1 2 4 5 6; do echo "ABLATE=$a"; POLUS_GEMM_P=2 POLUS_GEMM_ABLATE=$a timeout -k 10 120 python tools/gemm_bench.py --only ffn1,ffn2 2>&1 | grep -E " plain" | cut -c1-62; done
