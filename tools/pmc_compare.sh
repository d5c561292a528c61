#!/bin/bash
# PMC comparison of builds of the fused kernel (memory-path counters).  Usage: tools/pmc_compare.sh <lib.so|default[:variant]> ...
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
)
for SPEC in "$@"; do
  LIB=${SPEC%%:*}; VAR=${SPEC#*:}; [ "$VAR" == "$SPEC" ] && VAR=8
  TAG=$(basename $LIB .so)_v$VAR
  if [ "$LIB" != "default" ]; then export MELSPEC_LIB=$GRAFT_REPO_ROOT/mel_spec_amd/$LIB; else unset MELSPEC_LIB; fi
  export MELSPEC_VARIANT=$VAR
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1)); OUT=/tmp/pmc_${TAG}_$i; rm -rf $OUT
    timeout 100 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-host-io --no-traffic > $OUT.log 2>&1; echo "pass $i rc=$? $(date +%T)"
  done
  python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "whisper400" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", tag)
for k in sorted(acc):
    v = acc[k][len(acc[k]) // 2:]
    print(f"  {k:42s} {sum(v) / len(v):16.1f}")
PY
done
