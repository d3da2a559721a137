# Staggered start of the persistent ping-pong GEMM workgroups (POLUS_GEMM_STAGGER_US, default 6): the two launches it applies to
# from cold caches, then the step, interleaved on one box.  -> profiles/r03_ab_stagger.txt
cd $GRAFT_REPO_ROOT
echo "# python tools/pp_bench.py --cold --rounds 3 --iters 10 --ab POLUS_GEMM_STAGGER_US=0,2,4,6,8,12"
python3 tools/pp_bench.py --cold --rounds 3 --iters 10 --ab POLUS_GEMM_STAGGER_US=0,2,4,6,8,12 2>&1 | grep -E "gemm |ffn1|du "
echo "# bench.py --steps 30 --warmup 5, alternating POLUS_GEMM_STAGGER_US: ms/step"
for r in 1 2 3 4; do for v in 0 6; do POLUS_GEMM_STAGGER_US=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('POLUS_GEMM_STAGGER_US=$v', d['ms_per_step'])"; done; done
