#!/bin/bash
# one GPU call: fbank A/B (clip kernel vs two kernels) and the NeMo store sync sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_a.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
timeout 120 python tools/fbank_probe.py >> $O 2>&1
MELSPEC_LIB=$LAB MELSPEC_FB_CLIP=0 timeout 120 python tools/fbank_probe.py >> $O 2>&1
MELSPEC_LIB=$LAB MELSPEC_FB_CLIP=1 timeout 120 python tools/fbank_probe.py >> $O 2>&1
timeout 120 python tools/fbank_probe.py 2048 >> $O 2>&1
for m in 0 1 2 4 8 18 20 24; do
  echo "== MELSPEC_MM_SYNC=$m" >> $O
  MELSPEC_LIB=$LAB MELSPEC_MM_SYNC=$m timeout 200 python tools/nemo_probe.py >> $O 2>&1
done
cat $O
