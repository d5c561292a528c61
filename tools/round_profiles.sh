#!/bin/bash
# End-of-round evidence, one box: bench line, rocprofv3 stats + PMC of the bench kernel, the other configs, the guard bench, the aux
# kernels, the pow2 geometries, the vote's cost, kernel resources.  BEFORE it: build the previous round's library (git worktree + the two
# hipcc commands of tools/ab_build.sh -> mel_spec_amd/ab/lib_rNN.so) and run every tools/ab_run.py case against it (MELSPEC_LIB_OLDER=1):
# round 4 found a 29 % scheduling regression of the layout kernels that way (profiles/r04_sched_flip.txt).  Usage (GPU box): tools/round_profiles.sh r03   ->  gpurun_out/round_<tag>/*  (copy what is to be judged into profiles/)
TAG=${1:-r03}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/round_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
python - $OUT/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench:", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "speech ms", d["config"].get("speech", {}).get("ms"))
PY
# kernel statistics of the bench command itself (the timed loop's launches dominate the average)
PROFILE_CMD="python bench.py --steps 800 --warmup 100 --no-cpu-baseline --no-host-io --no-traffic --no-speech --no-legs" tools/profile.sh $TAG > $OUT/rocprof_summary.txt 2>&1
tools/profile_configs.sh $TAG > $OUT/rocprof_configs.txt 2>&1
python tools/guard_bench.py > $OUT/guard.txt 2>&1
tools/profile_aux.sh $TAG > /dev/null 2>&1
cp gpurun_out/aux_$TAG/table.txt $OUT/aux_table.txt; cp gpurun_out/aux_$TAG/summary.txt $OUT/aux_summary.txt
python tools/pow2_bench.py > $OUT/pow2.txt 2>&1
python tools/vote_cost.py > $OUT/vote_cost.txt 2>&1
python tools/kernel_resources.py > $OUT/kernel_resources.txt 2>&1
head -5 $OUT/rocprof_summary.txt; grep -A3 "whisper400_six_runs" $OUT/rocprof_summary.txt | grep -E "FETCH|WRITE|BANK|IDX_ACTIVE" | head
tail -26 $OUT/guard.txt | cut -c1-130
