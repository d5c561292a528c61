#!/bin/bash
# Effective shader clock of the fused kernel under different builds: GRBM_GUI_ACTIVE cycles (summed over the 8 XCDs) / duration.
# Usage: tools/clock_probe.sh <lib.so> ...   ("default" = the shipped library)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for LIB in "$@"; do
  OUT=/tmp/clk_$(basename $LIB .so)
  rm -rf $OUT
  if [ "$LIB" != "default" ]; then export MELSPEC_LIB=$GRAFT_REPO_ROOT/mel_spec_amd/$LIB; else unset MELSPEC_LIB; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-io --no-traffic > $OUT.log 2>&1
  python - "$OUT" "$LIB" <<'PY'
import csv, glob, sys
out, lib = sys.argv[1], sys.argv[2]
cyc, dur = [], {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "whisper400" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "GRBM_GUI_ACTIVE":
            cyc.append((r["Dispatch_Id"], float(r["Counter_Value"])))
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "whisper400" in r.get("Kernel_Name", ""):
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
pairs = [(c, dur[d]) for d, c in cyc if d in dur][50:]
if pairs:
    c = sum(p[0] for p in pairs) / len(pairs); t = sum(p[1] for p in pairs) / len(pairs)
    print(f"{lib}: n={len(pairs)} cycles(sum of 8 XCD)={c:.0f} duration={t/1e3:.1f} us -> {c/8/t:.3f} GHz", flush=True)
else:
    print(lib, "no data", len(cyc), len(dur))
PY
done
