#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small text summary (committed under profiles/)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("stats/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== PMC per dispatch (averaged over dispatches of each kernel) ==")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows("pmc*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "?")[:60]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
