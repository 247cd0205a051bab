set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03d; mkdir -p $O
pmc() { tag=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -- "$@" > /dev/null 2>&1; }
pmc fetch64 FETCH_SIZE -- python tools/run_batch.py 64 50000
pmc write64 WRITE_SIZE -- python tools/run_batch.py 64 50000
pmc lds64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -- python tools/run_batch.py 64 50000
python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("fetch64", "write64", "lds64"):
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "partial_dma_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[8:] or v
        out[k] = sum(v) / len(v)
json.dump(out, open("$O/pmc_batch_kernel_b64.json", "w"), indent=1)
print(out)
PY
rm -rf $O/pmc_fetch64 $O/pmc_write64 $O/pmc_lds64
