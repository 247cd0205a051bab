#!/bin/bash
# Run ON THE GPU BOX (via gpurun): HBM bytes fetched per 50k-patch bag by scores + pooling, one launch against two (FETCH_SIZE alone in
# its pass, per kernel, averaged over the launches after the first 12): gpurun_out/${VLSA_ROUND:-r06}/pmc_pool_traffic.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
N=${1:-50000}
for m in one two; do
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_pool_$m -- python tools/run_gated_pool.py $N $m > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {"N": $N, "algorithmic_bytes_X": $N * 1024}
for m in ("one", "two"):
    fs = glob.glob("$O/pmc_pool_%s/**/*counter_collection.csv" % m, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("vlsa::") and "prepare" not in k and r["Counter_Name"] == "FETCH_SIZE":
            acc[k].append(float(r["Counter_Value"]))
    per = {k: {"launches": len(v), "FETCH_SIZE_KB_avg": sum(v[12:]) / max(1, len(v[12:]))} for k, v in acc.items()}
    out[m] = per
    # FETCH_SIZE is in KB of 32-B units; gfx950 counts a 64-B request once -> x 2 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    out[m + "_fetched_bytes_per_bag_corrected"] = sum(d["FETCH_SIZE_KB_avg"] * d["launches"] for d in per.values()) / 36 * 1024 * 2
json.dump(out, open("$O/pmc_pool_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_pool_one $O/pmc_pool_two
