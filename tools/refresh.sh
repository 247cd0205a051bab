set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
(VLSA_GRAD_ERRORS_OUT=$O/grad_errors.txt timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_gpu.txt
VLSA_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_sharded_1rank.json 2> $O/bench_sh1.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 > $O/bench_4ranks_one_gpu.json 2> $O/bench_4ranks.err
VLSA_BENCH_RAMP=1 VLSA_BENCH_WATCHDOG=300 VLSA_BENCH_BACKEND=gloo timeout 500 python bench.py --gpus 8 --steps 2 --warmup 1 > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt | grep -v amdgpu
