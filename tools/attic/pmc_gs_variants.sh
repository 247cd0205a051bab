#!/bin/bash
# Run ON THE GPU BOX: shader cycles (not wall time: DVFS moves the clock with the data) of the ungated score kernel for library
# variants given as arguments (paths under vlsa_amd/_lib), 400k patches.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf $O/pmc_v_$tag
  VLSA_GS_HG2=1 VLSA_HIP_LIB=$PWD/vlsa_amd/_lib/$lib rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_v_$tag -- python tools/run_gated.py 400000 ungated > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_v_$tag/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "k_gated_scores" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
ks = glob.glob("$O/pmc_v_$tag/**/*kernel_trace.csv", recursive=True)
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ks[0])) if "k_gated_scores" in r["Kernel_Name"]][8:]
o = {k: sum(v[8:]) / len(v[8:]) for k, v in acc.items()}
us = sum(d) / len(d) / 1e3
cyc = o["GRBM_GUI_ACTIVE"] / 8
print(f"$tag: {us:7.1f} us  {cyc/1e3:7.1f} kcycles/XCD  {cyc/us/1e3:5.2f} GHz  wait_any {o['SQ_WAIT_ANY']/o['SQ_WAVE_CYCLES']:.2f} wait_inst {o['SQ_WAIT_INST_ANY']/o['SQ_WAVE_CYCLES']:.2f} active {o['SQ_ACTIVE_INST_ANY']/o['SQ_WAVE_CYCLES']:.2f}")
PY
done
