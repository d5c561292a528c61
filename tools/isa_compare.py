#!/usr/bin/env python3
"""Per-kernel comparison of two sets of gfx950 assembly files (hipcc --cuda-device-only -S): after a source move that must not change
any kernel, every kernel's instruction stream has to be identical.  tools/isa_compare.py OLD.s [OLD2.s ...] -- NEW.s [NEW2.s ...]
Local labels carry the function's index inside its translation unit (.LBB<idx>_<n>); they are normalised before comparing."""
import re, sys

def kernels(paths):
    out = {}
    for p in paths:
        name, body = None, []
        for line in open(p):
            m = re.match(r'^(_Z\w+):\s', line)
            if m:
                name, body = m.group(1), []
                continue
            if name is None:
                continue
            if line.startswith('.Lfunc_end'):
                out[name] = body
                name = None
                continue
            s = line.split(';')[0].rstrip()
            if not s or s.lstrip().startswith('.'):
                continue
            body.append(re.sub(r'\.LBB\d+_', '.LBB_', s))
    return out

i = sys.argv.index('--')
old, new = kernels(sys.argv[1:i]), kernels(sys.argv[i + 1:])
same = diff = 0
for k in sorted(set(old) | set(new)):
    if k not in old: print('only in NEW:', k); continue
    if k not in new: print('only in OLD:', k); continue
    if old[k] == new[k]: same += 1
    else:
        diff += 1
        print(f'DIFFERS ({len(old[k])} vs {len(new[k])} instructions): {k}')
print(f'{same} kernels identical, {diff} differ, {len(old)} old / {len(new)} new')
