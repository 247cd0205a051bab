# Same-box timing of k_tt_attn_bwd under launch-shape variations (-DVLSA_EXPERIMENT build): ticketed fold (VLSA_TT_NOSTATS) against the
# prefix-key workgroups, with the ablation bits of tools/attn_bwd_ablate.sh, the dynamic LDS size and the grid overridden (timing only).
O=gpurun_out/r06/attn_bwd_ablate2.txt; mkdir -p gpurun_out/r06; : > $O
E=$PWD/vlsa_amd/_lib/libvlsa_hip_exp.so
run() {   # NOSTATS ABL LDS GRID
  rm -rf gpurun_out/r06/abl
  ( [ "$1" = "1" ] && export VLSA_TT_NOSTATS=1; [ -n "$3" ] && export VLSA_TT_ATTN_LDS=$3; [ -n "$4" ] && export VLSA_TT_ATTN_GRID=$4;
    VLSA_HIP_LIB=$E VLSA_TT_ATTN_ABL=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/abl -- python tools/bench_text.py > /dev/null 2>&1 )
  echo "== NOSTATS=$1 ABL=$2 LDS=${3:-default} GRID=${4:-default}" >> $O
  python tools/kstats.py $(find gpurun_out/r06/abl -name "*kernel_stats.csv" | head -1) k_tt_attn_bwd >> $O
}
run 1 0 "" ""
run 1 1 "" ""
run 0 0 "" ""
run 0 16 "" ""
run 0 32 "" ""
run 0 18 "" ""
rm -rf gpurun_out/r06/abl; cat $O
