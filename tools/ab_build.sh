#!/bin/bash
# Build A/B variants of the library: tools/ab_build.sh name1:"-DA -DB" name2:"" ...  -> mel_spec_amd/ab/lib_<name>.so (parallel).
# Two translation units like mel_spec_amd/build.py: the flags after the colon go to both; RUNS_FLAGS (default: the shipped
# "-mllvm -amdgpu-sched-strategy=max-ilp") only to csrc/melspec_runs.hip.  RUNS_FLAGS=" " builds it with the default scheduler.
cd "$(dirname "$0")/.."
mkdir -p mel_spec_amd/ab
RUNS_FLAGS=${RUNS_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( d=$(mktemp -d /tmp/ab_XXXXXX)
    ( hipcc $COMMON $flags -c mel_spec_amd/csrc/melspec_hip.hip -o $d/a.o 2> mel_spec_amd/ab/$name.log ) &
    ( hipcc $COMMON $flags $RUNS_FLAGS -c mel_spec_amd/csrc/melspec_runs.hip -o $d/b.o 2> mel_spec_amd/ab/$name.runs.log ) &
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC -o mel_spec_amd/ab/lib_$name.so $d/a.o $d/b.o 2>> mel_spec_amd/ab/$name.log || echo "BUILD FAILED $name"
    rm -rf $d ) &
done
wait
ls -la mel_spec_amd/ab/*.so
