#!/bin/bash
# Build A/B variants of the library: tools/ab_build.sh name1:"-DA -DB" name2:"" ...  -> mel_spec_amd/ab/lib_<name>.so (parallel).
# One object per translation unit like mel_spec_amd/build.py (SOURCES / UNIT_FLAGS are read from it): the flags after the colon go to
# every unit; RUNS_FLAGS (default: the shipped "-mllvm -amdgpu-sched-strategy=max-ilp") only to csrc/melspec_runs.hip.
# RUNS_FLAGS=" " builds it with the default scheduler.  SRC_ROOT=<dir> builds another checkout's csrc (a previous round's library:
# git worktree add /tmp/prev <commit>; SRC_ROOT=/tmp/prev tools/ab_build.sh r05:"" -- its own build.py names its units).
cd "$(dirname "$0")/.."
OUT=$PWD/mel_spec_amd/ab
mkdir -p $OUT
SRC_ROOT=${SRC_ROOT:-$PWD}
RUNS_FLAGS=${RUNS_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function"
UNITS=$(cd $SRC_ROOT && python3 -c "from mel_spec_amd.build import SOURCES; print(' '.join(SOURCES))")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( d=$(mktemp -d /tmp/ab_XXXXXX)
    for u in $UNITS; do
      extra=""; [ $u = melspec_runs.hip ] && extra="$RUNS_FLAGS"
      ( hipcc $COMMON $flags $extra -c $SRC_ROOT/mel_spec_amd/csrc/$u -o $d/${u%.hip}.o 2> $OUT/$name.${u%.hip}.log ) &
    done
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $d/*.o 2>> $OUT/$name.log || echo "BUILD FAILED $name"
    rm -rf $d ) &
done
wait
ls -la $OUT/*.so
