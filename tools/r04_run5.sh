cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_text_tower.py tests/test_gpu_handler_loop.py -q -m gpu -x 2>&1 | tail -5) > $O/pytest_tt.txt
tail -3 $O/pytest_tt.txt
timeout 300 python tools/bench_text.py 2>&1 | grep -v amdgpu
VLSA_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_sharded_1rank.json 2> $O/bench_sh1.err; head -c 700 $O/bench_sharded_1rank.json; tail -2 $O/bench_sh1.err
