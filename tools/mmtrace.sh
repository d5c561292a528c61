cd /tmp; export TMPDIR=/tmp MELSPEC_LIB_OLDER=1
for l in r03 cur; do
  rm -rf /tmp/tr_$l
  MELSPEC_PRECISE=f MELSPEC_LIB=$GRAFT_REPO_ROOT/mel_spec_amd/ab/lib_$l.so rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$l -- python $GRAFT_REPO_ROOT/tools/layout_bench.py > /dev/null 2>&1
  f=$(find /tmp/tr_$l -name "*kernel_trace.csv" | head -1)
  python3 - $f $l <<'PY'
import csv, sys, statistics
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'six_kernel' in r['Kernel_Name']]
print(sys.argv[2], len(rows), 'keys', [k for k in rows[0].keys()][:30])
r=rows[len(rows)//2]
print({k:r[k] for k in r if k not in ('Kernel_Name',)})
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
print('median us', statistics.median(d)/1e3)
PY
done
