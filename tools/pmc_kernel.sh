#!/bin/bash
# SQ / LDS counters of the kernels of a command, two rocprofv3 passes (counters only, --kernel-trace): tools/pmc_kernel.sh <tag> <cmd...>
#   -> gpurun_out/pmck_<tag>.txt: per kernel the averages over its launches (second half of them: past the clock ramp)
TAG=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); OUT=/tmp/pmck_${TAG}_$i; rm -rf $OUT
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1; echo "pass $i rc=$?"
done
python - "$TAG" > gpurun_out/pmck_$TAG.txt <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmck_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x[len(x) // 2:]) / len(x[len(x) // 2:]) for c, x in v.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    wc = m["SQ_WAVE_CYCLES"]
    print("==", k)
    for c in sorted(m): print(f"   {c:24s} {m[c]:16.4g}   {m[c] / wc:8.4f} of wave cycles")
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        print(f"   LDS conflict / active {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f};  LDS active per CU / kernel cycles {m['SQ_LDS_IDX_ACTIVE'] / 256 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}"
              f";  VALU busy per SIMD / kernel cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}")
PY
cat gpurun_out/pmck_$TAG.txt
