cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_gated_scores.py -q -m gpu -x 2>&1 | tail -3) ; NS=400000,50000,100000,200000 timeout 1500 python tools/kbench_gated_ab.py - VLSA_GS_ROWSK=0 2>&1 | grep gated= 
