#!/bin/bash
# one GPU call: fbank clip kernel ablation, normaliser knobs, WRITE/FETCH_SIZE of the NeMo store modes and of the fbank paths
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/probe_b.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
echo "== fbank: product / no subtract / two kernels" >> $O
timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
MELSPEC_LIB=$LAB MELSPEC_FB_CLIP_SKIP=1 timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
MELSPEC_LIB=$LAB MELSPEC_FB_CLIP=0 timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
echo "== normaliser: fold wave selection" >> $O
for f in -1 3 5 8; do
  echo "MELSPEC_NORM_FOLD=$f" >> $O
  NEMO_ONLY=norm MELSPEC_MM_SYNC=0 MELSPEC_LIB=$LAB MELSPEC_NORM_FOLD=$f timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels >> $O
done
echo "== normaliser: LDS per workgroup x workgroups per CU (fold 8)" >> $O
for kp in "25 6" "50 3" "76 2" "19 8"; do
  set -- $kp
  echo "MELSPEC_NORM_KB=$1 MELSPEC_NORM_PER_CU=$2" >> $O
  NEMO_ONLY=norm MELSPEC_MM_SYNC=0 MELSPEC_LIB=$LAB MELSPEC_NORM_FOLD=8 MELSPEC_NORM_KB=$1 MELSPEC_NORM_PER_CU=$2 timeout 200 python tools/nemo_probe.py 2>&1 | grep normalize >> $O
done
echo "== WRITE_SIZE / FETCH_SIZE (KiB per launch)" >> $O
for cfg in "0 1" "2 1" "18 1" "0 0"; do
  set -- $cfg
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/nw
    MELSPEC_LIB=$LAB MELSPEC_MM_SYNC=$1 MELSPEC_FB_CLIP=$2 timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/nw -- python tools/nemo_write_probe.py > /tmp/nw.log 2>&1
    python - $1 $2 $c >> $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/nw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "synth" in k or "copyBuffer" in k: continue
        acc[k[:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"MM_SYNC={sys.argv[1]} FB_CLIP={sys.argv[2]} {sys.argv[3]} n={len(v)} mean={sum(v)/len(v):.0f} KiB = {sum(v)/len(v)*1024/1e6:.1f} MB  {k}")
PY
  done
done
cat $O
