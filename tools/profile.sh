#!/bin/bash
# Collect rocprofv3 kernel stats + PMC passes for the bench workload. Usage: tools/profile.sh <tag>
# (PMC in their own runs with --kernel-trace only, as the pool requires.)
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD=${PROFILE_CMD:-"python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-traffic --no-speech --no-legs"}
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
