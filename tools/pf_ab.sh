# Same-box A/B of the text tower's weight prefetch (prefetch_next, text_tower.hip): the shipped library against a -DVLSA_EXPERIMENT build
# (vlsa_amd/_lib/libvlsa_hip_exp.so: VLSA_EXTRA_HIPCC_FLAGS=-DVLSA_EXPERIMENT python -m vlsa_amd.build, copied) with the prefetch switched
# off (VLSA_TT_NOPF) and with every block reading block 0's weights (VLSA_TT_SAMEW: the cache-resident bound; results meaningless).
mkdir -p gpurun_out/r06
O=gpurun_out/r06/pf_ab.txt
E=$PWD/vlsa_amd/_lib/libvlsa_hip_exp.so
: > $O
for i in 1 2; do
echo "== default lib (prefetch)" >> $O; python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, VLSA_TT_NOPF (no prefetch)" >> $O; VLSA_HIP_LIB=$E VLSA_TT_NOPF=1 python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, prefetch on" >> $O; VLSA_HIP_LIB=$E python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, SAMEW + NOPF (weights cache-resident)" >> $O; VLSA_HIP_LIB=$E VLSA_TT_NOPF=1 VLSA_TT_SAMEW=1 python tools/bench_text.py 2>&1 | grep GPU >> $O
echo "== exp lib, SAMEW + prefetch" >> $O; VLSA_HIP_LIB=$E VLSA_TT_SAMEW=1 python tools/bench_text.py 2>&1 | grep GPU >> $O
done
echo "== train step, default lib" >> $O; (VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py tcga 30 2>&1 | grep ms_per_step | tail -1) >> $O
echo "== train step, NOPF" >> $O; (VLSA_HIP_LIB=$E VLSA_TT_NOPF=1 VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py tcga 30 2>&1 | grep ms_per_step | tail -1) >> $O
cat $O
