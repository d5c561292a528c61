#!/bin/bash
# same-box A/B (boxes differ by up to 10 %): a previous commit's library against the current one.  Build the former first:
#   git worktree add /tmp/prev <commit> && SRC_ROOT=/tmp/prev tools/ab_build.sh prev:"-DMELSPEC_LAB" &&
#   cp mel_spec_amd/ab/lib_prev.so mel_spec_amd/libmelspec_hip_prev.so
# then: gpurun -- tools/ab_same_box.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_g.txt
: > $O
PREV=$PWD/mel_spec_amd/libmelspec_hip_prev.so
for rep in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MELSPEC_LIB=$PREV; else unset MELSPEC_LIB; fi
    echo "== $lib (rep $rep)" >> $O
    timeout 120 python tools/fbank_probe.py 2>&1 | grep fbank >> $O
    NEMO_ONLY=norm timeout 200 python tools/nemo_probe.py 2>&1 | grep n_mels >> $O
    timeout 100 python tools/w512_bench.py 2>&1 | grep fused >> $O
  done
done
unset MELSPEC_LIB
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
cat $O
