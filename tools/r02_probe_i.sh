#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_i.txt
: > $O
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
echo "== normaliser stagger modes (0 none, 1 short first round, 2 sleep by (b>>8)&3, 3 sleep by hash)" >> $O
for rep in 1 2; do for st in 0 1 2 3; do
  echo "MELSPEC_NORM_STAGGER=$st" >> $O
  NEMO_ONLY=norm MELSPEC_LIB=$LAB MELSPEC_NORM_STAGGER=$st timeout 200 python tools/nemo_probe.py 2>&1 | grep normalize >> $O
done; done
echo "== product" >> $O
timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels >> $O
timeout 100 python tools/w512_bench.py 2>&1 | grep fused >> $O
cat $O
