#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_h.txt
: > $O
PREV=$PWD/mel_spec_amd/libmelspec_hip_prev.so
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
BP=$PWD/mel_spec_amd/libmelspec_hip_bperm.so
for rep in 1 2; do
  for lib in prev new bperm; do
    if [ $lib = prev ]; then export MELSPEC_LIB=$PREV; elif [ $lib = new ]; then export MELSPEC_LIB=$LAB; else export MELSPEC_LIB=$BP; fi
    echo "== $lib (rep $rep): fbank clip kernel / fused kernel + cmn_kernel / nemo / w512" >> $O
    timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
    MELSPEC_FB_CLIP=0 timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
    NEMO_ONLY=norm MELSPEC_MM_SYNC=0 timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels | head -1 >> $O
    timeout 100 python tools/w512_bench.py 2>&1 | grep fused >> $O
  done
done
export MELSPEC_LIB=$LAB
for m in 0 2 18 0 2 18; do
  echo "== MELSPEC_MM_SYNC=$m" >> $O
  NEMO_ONLY=norm MELSPEC_MM_SYNC=$m timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels >> $O
done
unset MELSPEC_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fbank" 2>&1 | tail -3 >> $O
cat $O
