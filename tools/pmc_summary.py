"""Average rocprofv3 --pmc counters per kernel name (substring filters given on the command line)."""
import collections, csv, glob, sys
root, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for p in pats:
            if p in name:
                acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
for p, cs in acc.items():
    print(p, {k: round(sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:])), 1) for k, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
