cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_handler_loop.py tests/test_ingest.py tests/test_gpu_modules.py tests/test_gpu_deepmil_batch.py tests/test_gpu_zeroshot_big.py -q -m gpu -x 2>&1 | tail -30) > $O/pytest_part.txt
tail -5 $O/pytest_part.txt
python tools/bench_module.py 2>&1 | grep -v amdgpu | tail -3 > $O/bench_module_la.txt; cat $O/bench_module_la.txt
for B in 4 8 16 32 64; do python tools/bench_attn.py 50000 bf16 $B 2>&1 | grep want_attn | sed "s/^/B=$B /"; done > $O/bench_attn_bpl.txt; cat $O/bench_attn_bpl.txt
