#!/bin/bash
# Build A/B variants of the library: tools/ab_build.sh name1:"-DA -DB" name2:"" ...  -> mel_spec_amd/ab/lib_<name>.so (parallel)
cd "$(dirname "$0")/.."
mkdir -p mel_spec_amd/ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-function $flags \
      -o mel_spec_amd/ab/lib_$name.so mel_spec_amd/csrc/melspec_hip.hip 2> mel_spec_amd/ab/$name.log || echo "BUILD FAILED $name" ) &
done
wait
ls -la mel_spec_amd/ab/*.so
