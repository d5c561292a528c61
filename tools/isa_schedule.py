#!/usr/bin/env python3
"""One character per instruction of a kernel's ISA (G global load, S global store, r/w LDS read/write, . VALU, s SALU, b branch, P s_setprio,
[vN lM] s_waitcnt): makes serialised memory / LDS round trips visible at a glance.  Usage: tools/isa_schedule.py <substring of the mangled name> [...]"""
import os, re, subprocess, sys, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/melspec_isa.s"
sys.path.insert(0, ROOT)
from mel_spec_amd.build import SOURCES, UNIT_FLAGS          # every translation unit with the flags the library build gives it
s = ""
for src in SOURCES:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "mel_spec_amd", "csrc", src), "-o", asm] + UNIT_FLAGS.get(src, []), check=True, stderr=subprocess.DEVNULL, cwd="/tmp")
    s += open(asm).read()
def cls(l):
    op = l.split()[0]
    if op.endswith(':'): return '|'
    if op.startswith('global_load'): return 'G'
    if op.startswith('global_store'): return 'S'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_write'): return 'w'
    if op.startswith('s_waitcnt'):
        m = re.search(r'vmcnt\((\d+)\)', l); n = re.search(r'lgkmcnt\((\d+)\)', l)
        return '[' + ('v' + m.group(1) if m else '') + ('l' + n.group(1) if n else '') + ']'
    if op.startswith('v_'): return '.'
    if op.startswith('s_setprio'): return 'P'
    if 'branch' in op: return 'b'
    return 's'
for name in re.findall(r'^(_ZN7melspec\S+):\s*; @', s, re.M):
    if not all(a in name for a in sys.argv[1:]): continue
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    body = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
    print(name)
    print('\n'.join(textwrap.wrap(''.join(cls(l) for l in body), 160)))
