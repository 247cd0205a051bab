# Timing-only ablations of the tower's attention backward (k_tt_attn_bwd) inside the tower's forward + backward, same box:
# -DVLSA_EXPERIMENT build (vlsa_amd/_lib/libvlsa_hip_exp.so), VLSA_TT_ATTN_ABL bits: 1 = no fold of the prefix rows' dK / dV (no ticket),
# 2 = no phase 2 (dQ / dK / dV), 4 = no phase 1 (scores), 8 = loads only.  Results are wrong with any bit set.
mkdir -p gpurun_out/r06
O=gpurun_out/r06/attn_bwd_ablate.txt
E=$PWD/vlsa_amd/_lib/libvlsa_hip_exp.so
: > $O
for abl in ${ABLS:-0 1 3 7 8}; do
  rm -rf gpurun_out/r06/abl
  VLSA_HIP_LIB=$E VLSA_TT_ATTN_ABL=$abl rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/abl -- python tools/bench_text.py > /dev/null 2>&1
  echo "== VLSA_TT_ATTN_ABL=$abl" >> $O
  python tools/kstats.py $(find gpurun_out/r06/abl -name "*kernel_stats.csv" | head -1) k_tt_attn_bwd >> $O
done
rm -rf gpurun_out/r06/abl
cat $O
