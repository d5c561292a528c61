#!/bin/bash
# SQ-level view of the f64 kernels (tools/aux_bench.py flavours): where the wave-cycles go.  Usage (GPU box): tools/pmc_flavours.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AUX_ITERS=6
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pf$i
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pf$i -- python tools/aux_bench.py flavours > /tmp/pf$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pf*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "melspec" not in k or "synth" in k: continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   wave-cycles %.3g: active %.0f %%  wait_any %.0f %%  wait_inst %.0f %% (lds %.0f %%) | VALU insts %.3g  LDS insts %.3g  LDS idx active %.3g (conflict %.0f %%)" % (
        wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
        m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_LDS_IDX_ACTIVE", 0), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
