#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_c.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
echo "== fbank: product (subtraction inside the unit loop) / no subtract / two kernels" >> $O
timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
MELSPEC_LIB=$LAB MELSPEC_FB_CLIP_SKIP=1 timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
timeout 120 python tools/fbank_probe.py 2048 2>&1 | grep fbank >> $O
timeout 120 python tools/fbank_probe.py 512 2>&1 | grep fbank >> $O
echo "== normaliser ablations (MELSPEC_NORM_SKIP: 1 no folds, 2 no stores, 4 no loads)" >> $O
for k in 0 1 2 4 3 5 6 7; do
  echo "MELSPEC_NORM_SKIP=$k" >> $O
  NEMO_ONLY=norm MELSPEC_LIB=$LAB MELSPEC_NORM_SKIP=$k timeout 200 python tools/nemo_probe.py 2>&1 | grep normalize >> $O
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fbank or nemo" 2>&1 | tail -5 >> $O
cat $O
