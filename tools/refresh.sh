# The short refresh after a change (ON THE GPU BOX, via gpurun): GPU suite, bench lines (default + driver arguments) + the kernel stats of
# the bench command, the N > 1 runs with 1 RCCL rank / 2-8 ranks sharing the GPU, the training-step and text-side files, smoke().
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
(VLSA_GRAD_ERRORS_OUT=$O/grad_errors.txt timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --streams 1 > $O/bench_profiled_streams1.json 2>/dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/stats
(VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py both 30) > $O/bench_train_step_graph.txt 2>&1
(VLSA_BENCH_TRAIN_MODE=eager python tools/bench_train_step.py both 30) > $O/bench_train_step_eager.txt 2>&1
VLSA_BENCH_TRAIN_MODE=graph rocprofv3 --kernel-trace --output-format csv -d $O/prof_step -- python tools/bench_train_step.py tcga 60 > /dev/null 2>&1
python tools/step_kernels.py $O/prof_step > $O/step_kernels.txt 2>&1; rm -rf $O/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $O/text -- python tools/bench_text.py > /dev/null 2>&1
cp $(find $O/text -name "*kernel_stats.csv" | head -1) $O/text_kernel_stats.csv; rm -rf $O/text
python tools/bench_text.py --cpu > $O/bench_text.txt 2>&1
python tools/bench_step.py > $O/bench_step.txt 2>&1
python tools/prof_single_slide.py 2>&1 | grep "N=" > $O/single_slide.txt
VLSA_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_sharded_1rank.json 2> $O/bench_sh1.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 > $O/bench_4ranks_one_gpu.json 2> $O/bench_4ranks.err
VLSA_BENCH_RAMP=1 VLSA_BENCH_WATCHDOG=300 VLSA_BENCH_BACKEND=gloo timeout 500 python bench.py --gpus 8 --steps 2 --warmup 1 > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt | grep -v amdgpu
