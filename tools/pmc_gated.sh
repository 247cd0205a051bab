#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC passes of the fragment-order attention-score kernel k_gated_scores alone (gated and ungated
# module, N patches per bag; VLSA_GS_TILE=0: the persistent LDS-DMA kernel has its own script, tools/pmc_tile.sh):
# gpurun_out/${VLSA_ROUND:-r06}/pmc_scores_<gated|ungated>_<N>.json
export VLSA_GS_TILE=0
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
N=${1:-393216}     # 12 whole rounds of the gated kernel: one launch per call
pmc() { tag=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -- "$@" > /dev/null 2>&1; }
for m in gated ungated; do
  pmc gs_${m}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -- python tools/run_gated.py $N $m
  pmc gs_${m}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE -- python tools/run_gated.py $N $m
  pmc gs_${m}_mem FETCH_SIZE -- python tools/run_gated.py $N $m
  pmc gs_${m}_wait SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -- python tools/run_gated.py $N $m
  # round 5: is the L2 -> CU path the limiter?  vector-memory requests of the CUs into L2, L2 requests / hits / misses, the time the
  # texture path is busy or stalled (one or two counters per pass: a name this driver does not know only loses its own pass)
  pmc gs_${m}_l2a TCP_TCC_READ_REQ_sum TCC_REQ_sum -- python tools/run_gated.py $N $m
  pmc gs_${m}_l2b TCC_HIT_sum TCC_MISS_sum -- python tools/run_gated.py $N $m
  pmc gs_${m}_l2c TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -- python tools/run_gated.py $N $m
  pmc gs_${m}_l2d TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum -- python tools/run_gated.py $N $m
  pmc gs_${m}_l2e TA_TA_BUSY_sum TCC_BUSY_sum -- python tools/run_gated.py $N $m
  python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("gs_${m}_sq", "gs_${m}_lds", "gs_${m}_mem", "gs_${m}_wait", "gs_${m}_l2a", "gs_${m}_l2b", "gs_${m}_l2c", "gs_${m}_l2d", "gs_${m}_l2e"):
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_gated_scores" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[8:] or v
        out[k] = sum(v) / len(v)
json.dump(out, open("$O/pmc_scores_${m}_$N.json", "w"), indent=1)
print("$m", json.dumps(out))
PY
  rm -rf $O/pmc_gs_${m}_*
done
