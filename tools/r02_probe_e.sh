#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_e.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fbank or nemo" 2>&1 | tail -5 >> $O
echo "== fbank product" >> $O
timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
echo "== normaliser (granule-aligned staging at any row alignment)" >> $O
NEMO_ONLY=norm timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels >> $O
for kp in "38 4" "30 5" "25 6" "50 3"; do
  set -- $kp
  echo "MELSPEC_NORM_KB=$1 MELSPEC_NORM_PER_CU=$2" >> $O
  NEMO_ONLY=norm MELSPEC_LIB=$LAB MELSPEC_NORM_KB=$1 MELSPEC_NORM_PER_CU=$2 timeout 200 python tools/nemo_probe.py 2>&1 | grep normalize >> $O
done
for k in 1 2 4 7; do
  echo "MELSPEC_NORM_SKIP=$k" >> $O
  NEMO_ONLY=norm MELSPEC_LIB=$LAB MELSPEC_NORM_SKIP=$k timeout 200 python tools/nemo_probe.py 2>&1 | grep normalize >> $O
done
cat $O
