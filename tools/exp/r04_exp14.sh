cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e14; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 1 0 1 0; do POLUS_ATTN_FWD_PERSIST=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu | sed "s/^/persist=$v /"; done > $O/attn_bench.txt; cat $O/attn_bench.txt
POLUS_ATTN_FWD_PERSIST=1 timeout -k 10 120 python3 tools/attn_bench.py --seq 128 --batch 32 2>&1 | grep -v amdgpu | sed "s/^/persist=1 /"; POLUS_ATTN_FWD_PERSIST=0 timeout -k 10 120 python3 tools/attn_bench.py --seq 128 --batch 32 2>&1 | grep -v amdgpu | sed "s/^/persist=0 /"
bash tools/ab_bench.sh POLUS_ATTN_FWD_PERSIST "0 1" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
