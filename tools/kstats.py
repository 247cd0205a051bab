"""Print (name, calls, avg us, min us) of a rocprofv3 kernel_stats.csv for kernels whose name contains one of the given substrings."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or any(p in r["Name"] for p in sys.argv[2:]):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f} us')
