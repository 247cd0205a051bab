#!/bin/bash
# Timing-only ablations of the attention-score kernel (VLSA_GS_ABL bits, vlsa_amd/csrc/gated_scores.hip): builds one library per
# variant HERE (CPU container), then `gpurun -- python tools/kbench_gated_ab.py - VLSA_HIP_LIB=... ...` times them on one box.
set -e
cd "$(dirname "$0")/.."
for b in ${@:-1 2 4 8 16 32 6 63}; do
    touch vlsa_amd/csrc/gated_scores.hip
    VLSA_EXTRA_HIPCC_FLAGS=-DVLSA_GS_ABL=$b python -c "import __graft_entry__ as g; g.build()" | tail -1
    cp vlsa_amd/_lib/libvlsa_hip.so vlsa_amd/_lib/libvlsa_hip_abl$b.so
done
touch vlsa_amd/csrc/gated_scores.hip
python -c "import __graft_entry__ as g; g.build()" | tail -1
