#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe_j.txt
: > $O
PREV=$PWD/mel_spec_amd/libmelspec_hip_prev.so
LAB=$PWD/mel_spec_amd/libmelspec_hip_lab.so
for rep in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MELSPEC_LIB=$PREV; else export MELSPEC_LIB=$LAB; fi
    echo "== $lib (rep $rep)" >> $O
    timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-host-io 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['roofline']['frac'])" >> $O
    timeout 300 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline --no-host-io 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['ms_per_step'], d['roofline']['frac'])" >> $O
  done
done
unset MELSPEC_LIB
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
timeout 100 python tools/layout_bench.py 2>&1 | tail -6 >> $O
cat $O
