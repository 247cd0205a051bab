#!/bin/bash
# Run ON THE GPU BOX (via gpurun): full GPU test suite, bench line, rocprofv3 kernel stats of the bench command,
# and separate PMC passes (HBM traffic) for the streaming kernel.  Outputs under gpurun_out/r01/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r01; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --streams 1 > $O/bench_streams1.json 2>/dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/run_batch.py 32 50000 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/run_batch.py 32 50000 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -- python tools/run_batch.py 32 50000 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_lds -- python tools/run_batch.py 32 50000 > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "partial_dma_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[8:] or v            # skip warm-up launches
        out[k] = sum(v) / len(v)
json.dump(out, open("$O/pmc_batch_kernel.json", "w"), indent=1)
print(out)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python tools/prof_train.py > /dev/null 2>&1
cp $(find $O/train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python tools/bench_train.py > $O/bench_train.txt 2>&1
python tools/bench_ingest.py > $O/bench_ingest.txt 2>&1
python tools/sweep_groups.py > $O/sweep_groups.txt 2>&1
python tools/kbench_gated.py > $O/kbench_gated.txt 2>&1
python tools/bench_deepmil.py > $O/bench_deepmil.txt 2>&1
python tools/bench_module.py > $O/bench_module.txt 2>&1
python tools/bench_paths.py > $O/bench_paths.txt 2>&1
python tools/bench_zeroshot.py > $O/bench_zeroshot.txt 2>&1
VLSA_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_sharded_1rank.json 2>/dev/null
rm -rf $O/train $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
cat $O/pytest_gpu.txt; cat $O/bench.json | cut -c1-600
