mkdir -p gpurun_out/r06
O=gpurun_out/r06/pf2_ab.txt
E=$PWD/vlsa_amd/_lib/libvlsa_hip_exp.so
: > $O
for i in 1 2 3; do
echo "== default lib (weights + regions)" >> $O; python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, VLSA_TT_NOPF2 (weights only)" >> $O; VLSA_HIP_LIB=$E VLSA_TT_NOPF2=1 python tools/bench_text.py 2>&1 | grep GPU >> $O
done
(timeout 600 python -m pytest tests/test_gpu_text_tower.py -x -q -m gpu 2>&1 | tail -1) >> $O
cat $O
