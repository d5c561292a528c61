#!/bin/bash
# rocprofv3 kernel statistics + HBM byte counters of the layout launches of a bank (tools/layout_bench.py: frame-major, mel-major, padded):
# tools/profile_layouts.sh <tag> <n_mels>   ->  gpurun_out/prof_<tag>/summary.txt   (round 6: the twelve-wave kernels of the 128- / 64- / 40-mel banks)
TAG=${1:-lay128}; export L_MELS=${2:-128}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="python tools/layout_bench.py"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
