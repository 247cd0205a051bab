#!/bin/bash
# Run ON THE GPU BOX: shader clock of k_scores_tile and of its timing-only ablations (GRBM_GUI_ACTIVE cycles / kernel duration)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
for v in 0 ${GT_ONLY:-6 22}; do
  lib=vlsa_amd/_lib/variants/libvlsa_gt$v.so; [ $v = 0 ] && lib=vlsa_amd/_lib/libvlsa_hip.so
  [ -f $lib ] || continue
  VLSA_HIP_LIB=$PWD/$lib rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/clk_$v -- python tools/run_gated.py 393216 gated > /dev/null 2>&1
  python - <<PY
import csv, glob
cs = glob.glob("$O/clk_$v/**/*counter_collection.csv", recursive=True)
ks = glob.glob("$O/clk_$v/**/*kernel_trace.csv", recursive=True)
cyc = [float(r["Counter_Value"]) for r in csv.DictReader(open(cs[0])) if "k_scores_tile" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
mf = [float(r["Counter_Value"]) for r in csv.DictReader(open(cs[0])) if "k_scores_tile" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(ks[0])) if "k_scores_tile" in r["Kernel_Name"]]
cyc, mf, dur = cyc[8:], mf[8:], dur[8:]
c = sum(cyc) / len(cyc) / 8; d = sum(dur) / len(dur)
print("ABL=$v: %.0f k cycles per XCD, %.1f us under the profiler -> %.2f GHz; MFMA busy %.1f %%" % (c / 1e3, d, c / d / 1e3, (sum(mf) / len(mf) / 1024 / c * 100) if mf else 0))
PY
  rm -rf $O/clk_$v
done
