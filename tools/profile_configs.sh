#!/bin/bash
# rocprofv3 kernel statistics + HBM byte counters of the other configs (tools/measure_configs.py: cfg3 fbank, cfg4 128 mels,
# cfg5 share, NeMo) and of the layout / Whisper-512 / precise kernels.  Usage: tools/profile_configs.sh <tag>
TAG=${1:-r01cfg}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="python tools/measure_configs.py cfg3 cfg4 cfg5 nemo"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python tools/layout_bench.py > $OUT/stats2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -- python tools/w512_bench.py > $OUT/stats3.log 2>&1
MELSPEC_PRECISE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python tools/measure_configs.py cfg2 > $OUT/stats4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
for d in stats2 stats3 stats4; do echo "== $d"; python - $OUT/$d <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**/*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print({k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "Percentage")})
PY
done >> $OUT/summary.txt
cat $OUT/summary.txt
