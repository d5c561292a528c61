#!/bin/bash
# LDS counters of a command, one rocprofv3 pass: tools/lds_pmc.sh <tag> <cmd...>  ->  gpurun_out/ldspmc_<tag>.txt
TAG=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ldspmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if m.get("SQ_LDS_IDX_ACTIVE", 0) == 0: continue
    print(k, " conflict/active %.3f  active/GRBM(per CU) %.3f  insts %.3g" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), m["SQ_LDS_IDX_ACTIVE"] / 256 / (m["GRBM_GUI_ACTIVE"] / 8), m["SQ_INSTS_LDS"]), {c: "%.3g" % x for c, x in m.items()})
PY
