#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv of tools/vote_cost.py and splits AUTO's calls into: the f32 kernel, the gap, the gated f64
kernel, the gap to the next call -- for light batches (the gated kernel returns at once) against calls without the vote."""
import csv, sys, statistics as st
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("six_runs", "wave_runs"):
        if k in n: return k
    if "precise" in n: return "precise2" if ", 2>" in n else "precise1"
    return "other"
seq = [(s, e, short(n)) for s, e, n in rows]
out = {}
for i in range(1, len(seq) - 2):
    s, e, k = seq[i]
    if k not in ("six_runs", "wave_runs"): continue
    nxt = seq[i + 1]
    prev = seq[i - 1]
    if nxt[2] == "precise2":
        light = (nxt[1] - nxt[0]) < 20000
        key = (k, "vote, light" if light else "vote, heavy")
        after = seq[i + 2]
        out.setdefault(key, []).append((e - s, nxt[0] - e, nxt[1] - nxt[0], after[0] - nxt[1], after[0] - s))
    elif nxt[2] == k and prev[2] == k:
        out.setdefault((k, "no vote"), []).append((e - s, 0, 0, nxt[0] - e, nxt[0] - s))
for key, v in sorted(out.items()):
    v = v[len(v) // 4:]            # drop the clock ramp
    med = [st.median(x[j] for x in v) / 1e3 for j in range(5)]
    print(f"{key[0]:10s} {key[1]:12s} n={len(v):5d}  f32 kernel {med[0]:8.2f} us  gap {med[1]:6.2f}  second kernel {med[2]:8.2f}  gap to next call {med[3]:6.2f}  call period {med[4]:8.2f} us")
