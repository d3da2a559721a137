# Round-3 A/B set (one box, interleaved): persistent ping-pong GEMM, tile order, side-stream dW, attention kernels.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; rm -rf $O; mkdir -p $O
{ echo "# python tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2  (warm operands)"; python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2 --rounds 5 --iters 10 2>&1 | grep -v amdgpu
  echo; echo "# ... --cold (512 MiB touched between launches: operands and outputs no longer cache-resident, as in the step)"; python3 tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2 --rounds 3 --iters 6 --cold 2>&1 | grep -v amdgpu
  echo; echo "# bench.py --steps 20 --warmup 5, alternating POLUS_GEMM_PERSIST (ms/step, us per K-contiguous GEMM launch)"
  for pv in 0 1 0 1 0 1; do POLUS_GEMM_PERSIST=$pv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('POLUS_GEMM_PERSIST=$pv', d['ms_per_step'], d['roofline']['avg_launch_us'])"; done; } > $O/ab_gemm_persistent.txt
{ echo "# python tools/pp_bench.py --ab POLUS_GEMM_ORDER=0,2,3,4,6 (column tiles an XCD's concurrent tiles span; 0 = row-major run), warm then --cold"; python3 tools/pp_bench.py --ab POLUS_GEMM_ORDER=0,2,3,4,6 --rounds 5 --iters 10 2>&1 | grep -v amdgpu; python3 tools/pp_bench.py --ab POLUS_GEMM_ORDER=0,2,3,4,6 --rounds 3 --iters 6 --cold 2>&1 | grep -v amdgpu; } > $O/ab_tile_order.txt
{ echo "# bench.py --config C --steps 10 --warmup 3, alternating POLUS_OVERLAP_DW (grouped dW on the side stream): samples/s, ms/step"
  for c in c2 c5 c3; do for ov in 1 0 1 0; do POLUS_OVERLAP_DW=$ov python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-loss100 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c POLUS_OVERLAP_DW=$ov', d['value'], d['ms_per_step'])"; done; done; } > $O/ab_overlap_dw.txt
{ python3 tools/attn_bench.py 2>&1 | grep -v amdgpu; python3 tools/attn_bench.py --seq 512 --batch 16 2>&1 | grep -v amdgpu; python3 tools/attn_bench.py --seq 128 --batch 32 2>&1 | grep -v amdgpu; } > $O/attn_bench.txt
python3 tools/debug/attn_bwd_ablate.py 2>&1 | grep -v amdgpu > $O/attn_bwd_kres_ablation.txt
cat $O/ab_gemm_persistent.txt | tail -8
