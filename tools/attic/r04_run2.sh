cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(VLSA_GRAD_ERRORS_OUT=$O/grad_errors.txt timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15) > $O/pytest_gpu.txt
grep -E "passed|failed" $O/pytest_gpu.txt
python tools/bench_module.py > $O/bench_module.txt 2>&1; cat $O/bench_module.txt | grep -v amdgpu
