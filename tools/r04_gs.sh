cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_text_tower.py tests/test_gpu_handler_loop.py tests/test_train_step.py tests/test_gpu_training_5fold.py tests/test_gpu_training_cindex.py tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -12)
python tools/bench_text.py 2>&1 | grep "GPU forward"
VLSA_TT_GRAPH=0 python tools/bench_text.py 2>&1 | grep "GPU forward" | sed 's/^/VLSA_TT_GRAPH=0: /'
python tools/bench_step.py 2>&1 | grep -v amdgpu | cut -c1-260
VLSA_TT_GRAPH=0 python tools/bench_step.py 2>&1 | grep -v amdgpu | grep "ms per optimizer" | cut -c1-200 | sed 's/^/VLSA_TT_GRAPH=0: /'
