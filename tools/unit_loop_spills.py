#!/usr/bin/env python3
"""Scratch (spill) instructions inside EVERY copy of a kernel's unit loop -- each span from an `s_setprio 0` to the first branch behind the
following `s_setprio 2` -- in a gfx950 assembly file.  tools/hotloop_spills.py looks at the first copy only; the run kernels have three
(the vote's loop, the static run, the handed-out chunks).  A scratch load inside a unit costs more than its latency: its s_waitcnt vmcnt(0)
also waits for the previous unit's stores.   tools/unit_loop_spills.py file.s [substring of the mangled name ...]"""
import re, sys
s = open(sys.argv[1]).read()
bad = 0
for name in re.findall(r'^(_ZN7melspec\S+):', s, re.M):
    if not all(a in name for a in sys.argv[2:]): continue
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    lines = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
    prio0 = [k for k, l in enumerate(lines) if l.startswith('s_setprio 0')]
    prio2 = [k for k, l in enumerate(lines) if l.startswith('s_setprio 2')]
    out = []
    for a in prio0:
        b = next((k for k in prio2 if k > a), None)
        if b is None: continue
        e = next((k for k in range(b, len(lines)) if lines[k].startswith(('s_cbranch', 's_branch'))), len(lines))
        n = sum('scratch_' in l for l in lines[a:e])
        bad += n
        out.append(f"[{a}..{e}: {n}]")
    if out: print(f"{len(lines):6d} instructions, unit-loop copies [first..last: scratch ops] {' '.join(out)}  {name[:90]}")
sys.exit(1 if bad else 0)
