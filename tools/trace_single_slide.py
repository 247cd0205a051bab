"""Kernel timeline of the one-slide-per-call loop from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/prof_single_slide.py 50k
    python tools/trace_single_slide.py <dir>
Prints, per kernel of the chain, its average duration and the average idle gap before it (end of the previous kernel -> its start),
over the last 600 calls of the trace."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
rows = rows[-1800:]
stat = {}
prev_end = None
for s, e, n in rows:
    st = stat.setdefault(n, [0, 0.0, 0.0])
    st[0] += 1
    st[1] += (e - s) / 1e3
    if prev_end is not None:
        st[2] += max(0, s - prev_end) / 1e3
    prev_end = e
tot = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{len(rows)} launches over {tot:.0f} us")
for n, (c, dur, gap) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:50s} calls {c:5d}  avg {dur / c:7.2f} us  idle before it {gap / c:6.2f} us")
