for a in 0 264 520 776 1032 1544; do echo "ABLATE=$a"; POLUS_GEMM_ABLATE=$a timeout -k 10 120 python tools/gemm_bench.py --only ffn1 2>&1 | grep -E "plain|bias\+gelu|gelu-bwd" | cut -c1-62; done
