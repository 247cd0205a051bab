# kernel stats of the batched optimizer step alone (tools/bench_step.py --only-batched): what runs next to the text tower
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python tools/bench_step.py --only-batched > /dev/null 2>&1
f=$(find /tmp/prof_step -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out/r04; cp $f gpurun_out/r04/step_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
steps=80.0
tot=sum(float(r["TotalDurationNs"]) for r in rows)
tt=sum(float(r["TotalDurationNs"]) for r in rows if "k_tt_" in r["Name"])
print("GPU us per step (2 bag-size sets x 40 steps): total", round(tot/steps/1e3,1), "tower", round(tt/steps/1e3,1))
for r in rows[:40]:
    if "k_tt_" in r["Name"]: continue
    print(r["Calls"], "us/step", round(float(r["TotalDurationNs"])/steps/1e3,1), "avg", round(float(r["AverageNs"])/1e3,1), r["Name"][:110])
PY
