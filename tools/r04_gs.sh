cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bagset.py tests/test_gpu_handler_loop.py tests/test_gpu_text_tower.py tests/test_gpu_batch.py tests/test_gpu_batch_backward.py tests/test_gpu_batch_attn.py tests/test_gpu_training_5fold.py -q -m gpu -x 2>&1 | tail -12)
python tools/bench_module.py 2>&1 | grep -v amdgpu > $O/bench_module.txt; cat $O/bench_module.txt
python tools/bench_text.py 2>&1 | grep "GPU forward"
python tools/bench_step.py 2>&1 | grep -v amdgpu
