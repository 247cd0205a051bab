cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_text_tower.py -q -m gpu -x 2>&1 | tail -3)
python tools/bench_text.py 2>&1 | grep "GPU forward"
VLSA_TT_ATTN_THREADS=256 python tools/bench_text.py 2>&1 | grep "GPU forward" | sed 's/^/attn 256 threads: /'
rocprofv3 --kernel-trace --stats --output-format csv -d $O/text2 -- python tools/bench_text.py > /dev/null 2>&1
cp $(find $O/text2 -name "*kernel_stats.csv" | head -1) $O/text_kernel_stats_b.csv; rm -rf $O/text2; head -14 $O/text_kernel_stats_b.csv | cut -c1-150
python tools/bench_module.py 2>&1 | grep "handler eval"
(VLSA_5FOLD_EPOCHS=10 VLSA_5FOLD_LR=2e-4 timeout 1500 python -m pytest tests/test_gpu_training_5fold.py -q -m gpu -x -s --durations=8 2>&1 | grep -E "fold|tcga|passed|failed|s call" | tail -20)
