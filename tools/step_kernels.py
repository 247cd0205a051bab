"""Per-step kernel list of the (graph-replayed) optimizer step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/bench_train_step.py tcga 60
    python tools/step_kernels.py <dir>
Steps are delimited by the loss kernel; prints kernels per step, span / busy / idle, and every kernel's share (10 steps before the last two)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
t = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(t)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_surv_loss" in r[2]]
a, b = idx[-12], idx[-2]
seg, n = rows[a:b], 10
tot = (seg[-1][0] - seg[0][0]) / 1e3 / n
busy = sum(e - s for s, e, _ in seg) / 1e3 / n
print(f"per step: {len(seg) / n:.1f} kernels, span {tot:.1f} us, busy {busy:.1f} us, idle {tot - busy:.1f} us")
agg = {}
for s, e, k in seg:
    k = k.split("(")[0]
    k = k[-70:]
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
for k, (c, tt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} x{c / n:5.1f}  {tt / n:8.1f} us/step  avg {tt / c:6.2f}")
if "--order" in sys.argv:       # the launches of ONE step in order (tower blocks folded: only the first and the last block are listed)
    one = rows[idx[-3]:idx[-2]]
    print("\none step, in launch order (start offset us, duration us, gap to the previous end us):")
    t0, prev = one[0][0], None
    for s, e, k in one:
        k = k.split("(")[0][-60:]
        print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.2f} {0.0 if prev is None else (s - prev) / 1e3:6.2f}  {k}")
        prev = e
