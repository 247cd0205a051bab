cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python tools/bench_step.py > /dev/null 2>&1
f=$(find /tmp/prof_step -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out/r04; cp $f gpurun_out/r04/step_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
tt=sum(float(r["TotalDurationNs"]) for r in rows if "k_tt_" in r["Name"])
print("total GPU ms", tot/1e6, "tower ms", tt/1e6)
for r in rows[:45]:
    if "k_tt_" in r["Name"]: continue
    print(r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,1), r["Name"][:130])
PY
