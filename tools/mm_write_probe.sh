#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the mel-major store under the MELSPEC_MM_SYNC modes (0 none, 1 workgroup barrier, 2/4/8 sub-group barrier)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in ${MODES:-0 1 2 4 8}; do
  rm -rf /tmp/mmw; 
  MELSPEC_MM_SYNC=$m L_REPS=20 timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/mmw -- python tools/layout_bench.py > /tmp/mmw.log 2>&1
  python - $m <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/mmw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "six_kernel" in r["Kernel_Name"]:
            acc["true" if "true>" in r["Kernel_Name"] else "false"].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("mode", sys.argv[1], "LAYOUT", k, "n", len(v), "WRITE_SIZE MB", sum(v) / len(v) * 1024 / 1e6)
PY
done
