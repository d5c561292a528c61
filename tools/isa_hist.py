#!/usr/bin/env python3
"""Instruction histogram of a kernel's unit loop, split at its s_setprio markers (phase 1 | phase 2 | phases 3-4), from an ISA dump:
tools/isa_hist.py <file.s> <substring of the mangled name>...   Classes: f64 VALU, f32/int VALU, DPP, cvt, LDS, global, SALU, waitcnt."""
import re, sys, collections
s = open(sys.argv[1]).read()
subs = sys.argv[2:]
for name in re.findall(r'^(_ZN7melspec\S+):\s*; @', s, re.M):
    if not all(a in name for a in subs): continue
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    lines = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
    # the unit loop: from the first "s_setprio 0" to the last "s_setprio 2" ... next backward branch
    prio = [k for k, l in enumerate(lines) if l.startswith('s_setprio')]
    if not prio: continue
    # take the LAST group of three prio markers (0,1,2) as the hot loop when several loops exist
    groups = []
    for k in prio:
        lvl = int(lines[k].split()[1])
        if lvl == 0: groups.append([k])
        elif groups: groups[-1].append(k)
    print(name[:120])
    for g in groups:
        if len(g) < 3: continue
        # end of phases 3-4: the next s_cbranch after g[2] that jumps backwards (approximate: next 's_cbranch' or 's_branch')
        end = next((k for k in range(g[2], len(lines)) if lines[k].startswith(('s_cbranch', 's_branch'))), len(lines))
        bounds = [g[0], g[1], g[2], end]
        tot = collections.Counter()
        for ph in range(3):
            c = collections.Counter()
            for l in lines[bounds[ph]:bounds[ph + 1]]:
                op = l.split()[0]
                if op.endswith(':'): continue
                if 'dpp' in l or op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'ds_bpermute', 'ds_swizzle')): k = 'dpp/lane'
                elif op.startswith('v_cvt'): k = 'cvt'
                elif op.startswith('v_') and ('_f64' in op or op in ('v_mov_b64',)): k = 'f64'
                elif op.startswith('v_pk_'): k = 'pk'
                elif op.startswith('v_'): k = 'valu32'
                elif op.startswith('ds_'): k = 'lds'
                elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): k = 'vmem' if not op.startswith('scratch') else 'scratch'
                elif op.startswith('s_waitcnt'): k = 'wait'
                elif op.startswith('s_'): k = 'salu'
                else: k = 'other'
                c[k] += 1
            tot.update(c)
            print(f"  phase {ph + 1 if ph < 2 else '3-4'}: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
        print("  total:   " + "  ".join(f"{k} {v}" for k, v in sorted(tot.items())) + f"   (lines {bounds[0]}..{bounds[3]})")
