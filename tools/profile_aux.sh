#!/bin/bash
# The aux kernels (tools/aux_bench.py): event-timed table, rocprofv3 kernel statistics, FETCH_SIZE / WRITE_SIZE per kernel.
# Usage (on the GPU box): tools/profile_aux.sh <tag>   ->  gpurun_out/aux_<tag>/{table.txt,summary.txt}
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/aux_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/aux_bench.py > $OUT/table.txt 2>&1
export AUX_ITERS=8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python tools/aux_bench.py > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc1 -- python tools/aux_bench.py stft quant vad flavours > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc2 -- python tools/aux_bench.py stft quant vad flavours > $OUT/pmc2.log 2>&1
python - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
print("rocprofv3 --kernel-trace --stats of tools/aux_bench.py (AUX_ITERS=8): kernel, calls, average / min ns")
for f in glob.glob(os.path.join(out, "stats", "**/*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {float(r['MinNs']):12.0f}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in (("pmc1", "FETCH_SIZE"), ("pmc2", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(out, d, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
print("\nHBM traffic per launch (KiB counters; bytes = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of profiles/r01_calibration.txt): kernel, launches, FETCH KiB, WRITE KiB, MB")
for k, v in sorted(acc.items()):
    fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    print(f"  {k[:100]:100s} {len(v['FETCH_SIZE']):4d} {fe:12.0f} {wr:12.0f} {(2 * fe + wr) * 1024 / 1e6:10.1f}")
PY
cat $OUT/table.txt; cat $OUT/summary.txt | cut -c1-200
