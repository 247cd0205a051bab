mkdir -p gpurun_out/r06
O=gpurun_out/r06/attn_stats_ab.txt
E=$PWD/vlsa_amd/_lib/libvlsa_hip_exp.so
(timeout 900 python -m pytest tests/test_gpu_text_tower.py tests/test_gpu_train_step_graph.py tests/test_train_step.py -x -q -m gpu 2>&1 | tail -2) > $O
for i in 1 2; do
echo "== default lib (prefix keys from the forward's statistics)" >> $O; python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, VLSA_TT_NOSTATS (ticketed fold)" >> $O; VLSA_HIP_LIB=$E VLSA_TT_NOSTATS=1 python tools/bench_text.py 2>&1 | grep GPU >> $O
done
rm -rf gpurun_out/r06/abl; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/abl -- python tools/bench_text.py > /dev/null 2>&1
python tools/kstats.py $(find gpurun_out/r06/abl -name "*kernel_stats.csv" | head -1) k_tt_attn >> $O; rm -rf gpurun_out/r06/abl
(VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py both 30 2>&1 | grep "\"ms_per_step" | head -2) >> $O
cat $O
