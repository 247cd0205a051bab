#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC passes of k_scores_tile alone (gated_scores_tile.hip): gpurun_out/${VLSA_ROUND:-r06}/pmc_tile_<gated|ungated>_<N>.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
N=${1:-393216}
MODS=${2:-gated ungated}
pmc() { tag=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -- "$@" > /dev/null 2>&1; }
for m in $MODS; do
  pmc gt_${m}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -- python tools/run_gated.py $N $m
  pmc gt_${m}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE -- python tools/run_gated.py $N $m
  pmc gt_${m}_mem FETCH_SIZE -- python tools/run_gated.py $N $m
  pmc gt_${m}_wait SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -- python tools/run_gated.py $N $m
  pmc gt_${m}_l2a TCP_TCC_READ_REQ_sum TCC_REQ_sum -- python tools/run_gated.py $N $m
  pmc gt_${m}_l2e TA_TA_BUSY_sum TCC_BUSY_sum -- python tools/run_gated.py $N $m
  python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("gt_${m}_sq", "gt_${m}_lds", "gt_${m}_mem", "gt_${m}_wait", "gt_${m}_l2a", "gt_${m}_l2e"):
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_scores_tile" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[8:] or v
        out[k] = sum(v) / len(v)
json.dump(out, open("$O/pmc_tile_${m}_$N.json", "w"), indent=1)
print("$m", json.dumps(out))
PY
  rm -rf $O/pmc_gt_${m}_*
done
