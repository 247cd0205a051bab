E=$PWD/vlsa_amd/_lib/libvlsa_hip_nont.so
mkdir -p gpurun_out/r06
for lib in default nont; do
  rm -rf gpurun_out/r06/p
  if [ $lib = nont ]; then export VLSA_HIP_LIB=$E; fi
  VLSA_BENCH_TRAIN_MODE=graph rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/p -- python tools/bench_train_step.py tcga 60 > /dev/null 2>&1
  echo "== $lib"; python tools/step_kernels.py gpurun_out/r06/p | grep -E "per step|dma_batch"
  VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py tcga 30 2>&1 | grep "\"ms_per_step" | head -1
done
rm -rf gpurun_out/r06/p
