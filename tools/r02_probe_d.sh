#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/probe_d.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
echo "== fbank clip kernel ablations (MELSPEC_FB_CLIP_SKIP: 1 none, 2 loads only, 4 stores only, 8 after the run)" >> $O
for k in 0 1 2 4 8 10 12; do
  MELSPEC_LIB=$LAB MELSPEC_FB_CLIP_SKIP=$k timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank | sed "s/^/SKIP=$k /" >> $O
done
echo "== normaliser: per-kernel durations (rocprofv3 --kernel-trace --stats)" >> $O
for k in 0 7; do
  rm -rf /tmp/np
  NEMO_ONLY=norm MELSPEC_LIB=$LAB MELSPEC_NORM_SKIP=$k timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python tools/nemo_probe.py > /tmp/np.log 2>&1
  echo "MELSPEC_NORM_SKIP=$k" >> $O
  python - >> $O <<'PY'
import csv, glob
for f in glob.glob("/tmp/np/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  ", r["Name"][:90], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
PY
done
cat $O
