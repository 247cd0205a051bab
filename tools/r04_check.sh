#!/bin/bash
# Round-4 quick check ON THE GPU BOX: GPU suite with the observed gradient errors recorded, the bench line (all legs), the
# self-launched 2-rank run of the N > 1 path with the ranks sharing the one GPU (gloo), smoke().
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(VLSA_GRAD_ERRORS_OUT=$O/grad_errors.txt timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15) > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-extra > $O/bench_2ranks_gloo_one_gpu.json 2> $O/bench_2ranks.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/pytest_gpu.txt; head -c 1500 $O/bench.json; echo; tail -3 $O/bench.err; head -c 600 $O/bench_2ranks_gloo_one_gpu.json; tail -3 $O/bench_2ranks.err; cat $O/smoke.txt | tail -1
