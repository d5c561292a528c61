#!/usr/bin/env python3
"""The instruction histogram of the kernel behind every leg of the bench line, for its `valu` fields (VERDICT r05 'missing' 4, 'next' 3):
compiles the library's units to gfx950 assembly with the flags mel_spec_amd/build.py gives them, takes each kernel's unit loop (between
its first `s_setprio 0` and the loop's back edge: tools/isa_hist.py's rule) and writes
  profiles/isa_hist.json   {source_hash, legs: {leg: {kernel, frames_per_unit, f64, valu32, cvt, dpp/lane, lds, vmem, salu, wait}}}
  stdout                   the table (profiles/r06_isa_hist.txt)
Static counts of ONE pass of the unit loop: branches inside it (clip edges, the guard's bookkeeping) are counted once, like isa_hist.py."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mel_spec_amd import build as B

# leg -> (unit, substrings the mangled name must contain, frames per unit)
LEGS = {
    "value":     ("melspec_runs.hip", ["whisper400_six_runs_kernel", "LensSixStaticILi80E"], 6),
    "cfg4":      ("melspec_runs.hip", ["whisper400_six_wide_runs_kernel"], 6),
    "mel_major": ("whisper400.hip", ["whisper400_six_kernelILi9E", "LensSixStaticILi80E"], 6),
    "f64":       ("whisper400.hip", ["whisper400_six64_kernelILi9E", "LensSixStaticILi80E"], 6),
    "speech":    ("whisper400.hip", ["whisper400_six64_kernelILi9E", "LensSixStaticILi80E"], 6),
    "speech128": ("whisper400.hip", ["whisper400_six64_kernelILi15E"], 6),
    # the compile-time banks are LensFbStatic<slot lengths>: Kaldi-80 <2,2,3,5,7,9>, Whisper-512's 80 mels <2,2,3,5,8,9>, NeMo-128 <1,1,1,2,2,3,4,5,7>
    "cfg3":      ("fbank512.hip", ["fbank512_clip_kernelILi6E", "LensFbStaticIJLi2ELi2ELi3ELi5ELi7ELi9E", "Lb0E"], 4),
    "w512":      ("fbank512.hip", ["w512_auto_kernelIfLi12ELi6E"], 4),          # default mode on noise-like input: the voting f32 launch
    "w512_f64":  ("fbank512.hip", ["fbank512_wave_kernelIdLi8ELi1ELi2ELi6E", "LensFbStaticIJLi2ELi2ELi3ELi5ELi8ELi9E", "Lb1E"], 4),
    "w512_f32":  ("fbank512.hip", ["fbank512_wave_kernelIfLi12ELi1ELi2ELi6E", "LensFbStaticIJLi2ELi2ELi3ELi5ELi8ELi9E", "Lb1E"], 4),
    "nemo":      ("fbank512.hip", ["fbank512_wave_kernelIdLi8ELi1ELi1ELi10E", "LensFbStaticIJLi1ELi1ELi1E", "Lb0E"], 4),
    "nemo_f32":  ("fbank512.hip", ["fbank512_wave_kernelIfLi12ELi1ELi1ELi10E", "LensFbStaticIJLi1ELi1ELi1E", "Lb0E"], 4),
}

def klass(l):
    op = l.split()[0]
    if 'dpp' in l or op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'ds_bpermute', 'ds_swizzle')): return 'dpp/lane'
    if op.startswith('v_cvt'): return 'cvt'
    if op.startswith('v_') and ('_f64' in op or op == 'v_mov_b64'): return 'f64'
    if op.startswith('v_pk_'): return 'pk'
    if op.startswith('v_'): return 'valu32'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'

def unit_loop(lines):
    """lines of the hot unit loop: from the last `s_setprio 0` that is followed by priorities 1 and 2 to the first branch behind `s_setprio 2`;
    kernels without the three markers: the whole body"""
    prio = [(k, int(l.split()[1])) for k, l in enumerate(lines) if l.startswith('s_setprio')]
    groups = []
    for k, lvl in prio:
        if lvl == 0: groups.append([k])
        elif groups: groups[-1].append(k)
    groups = [g for g in groups if len(g) >= 3]
    if not groups: return lines, False
    g = max(groups, key=lambda g: g[2] - g[0])          # the unit loop is the longest such span
    end = next((k for k in range(g[2], len(lines)) if lines[k].startswith(('s_cbranch', 's_branch'))), len(lines))
    return lines[g[0]:end], True

# legs whose unit loop holds code a pass does not execute (wave-uniform branches): the static count overstates them
NOTES = {
    "nemo": "phase 1 is in the loop three times (interior frames, interior without pre-emphasis, frames at a clip's ends: wave-uniform branches, "
            "one executes): the executed count is lower than this static one",
    "nemo_f32": "phase 1 is in the loop three times (interior frames, interior without pre-emphasis, frames at a clip's ends: wave-uniform branches, "
                "one executes): the executed count is lower than this static one",
}

asm = {}
with tempfile.TemporaryDirectory() as tmp:
    procs = {}
    for unit in sorted({u for u, _, _ in LEGS.values()} | {"melspec_runs.hip", "whisper400.hip", "fbank512.hip"}):
        out = os.path.join(tmp, unit + ".s")
        procs[unit] = (out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out,
                                             os.path.join(B.CSRC, unit)] + B.UNIT_FLAGS.get(unit, []), stderr=subprocess.DEVNULL))
    for unit, (out, p) in procs.items():
        if p.wait() != 0: raise SystemExit(f"hipcc failed on {unit}")
        asm[unit] = open(out).read()

res = {"source_hash": B.source_hash(), "note": "static instruction counts of one pass of each kernel's unit loop (tools/isa_legs.py)", "legs": {}}
print(f"# tools/isa_legs.py -- source hash {res['source_hash']}; wave-instructions of ONE pass of the unit loop, and per frame")
print(f"# {'leg':10s} {'fpu':>3s} {'f64':>5s} {'cvt':>4s} {'v32':>4s} {'dpp':>4s} {'lds':>4s} {'vmem':>4s} {'salu':>4s} {'wait':>4s} | per frame: f64, other VALU | kernel")
for leg, (unit, subs, fpu) in LEGS.items():
    s = asm[unit]
    names = [n for n in re.findall(r'^(_ZN7melspec\S+):\s*; @', s, re.M) if all(a in n for a in subs)]
    if len(names) != 1:
        print(f"# {leg}: {len(names)} kernels match {subs}", file=sys.stderr)
        continue
    name = names[0]
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    lines = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().split()[0].endswith(':')]
    body, marked = unit_loop(lines)
    c = collections.Counter(klass(l) for l in body)
    row = dict(c); row.update(kernel=name, frames_per_unit=fpu, unit_loop_marked=marked)
    if leg in NOTES: row["note"] = NOTES[leg]
    res["legs"][leg] = row
    other = c['valu32'] + c['cvt'] + c['dpp/lane'] + c['pk']
    print(f"  {leg:10s} {fpu:3d} {c['f64']:5d} {c['cvt']:4d} {c['valu32']:4d} {c['dpp/lane']:4d} {c['lds']:4d} {c['vmem']:4d} {c['salu']:4d} {c['wait']:4d} | {c['f64'] / fpu:7.1f} {other / fpu:7.1f} | "
          + subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110] + ("" if marked else "   [no priority markers: whole kernel]"))

# Scratch (spill) instructions inside any copy of a unit loop, for EVERY kernel of the compiled units that marks its phases (tools/unit_loop_spills.py's
# rule): a scratch reload inside a unit waits on vmcnt(0), i.e. also for the previous unit's stores -- round 6 found the 64-mel bank's kernels 19-29 %
# slower for six to eight of them.  tests/test_parallel.py holds the kernels the library dispatches to zero.
def loop_scratch(lines):
    prio0 = [k for k, l in enumerate(lines) if l.startswith('s_setprio 0')]
    prio2 = [k for k, l in enumerate(lines) if l.startswith('s_setprio 2')]
    total, copies = 0, 0
    for a in prio0:
        b = next((k for k in prio2 if k > a), None)
        if b is None: continue
        e = next((k for k in range(b, len(lines)) if lines[k].startswith(('s_cbranch', 's_branch'))), len(lines))
        total += sum('scratch_' in l for l in lines[a:e]); copies += 1
    return total if copies else None

res["unit_loop_scratch"] = {}
for unit, s in asm.items():
    for name in re.findall(r'^(_ZN7melspec\S+):\s*; @', s, re.M):
        i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
        lines = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
        n = loop_scratch(lines)
        if n is not None: res["unit_loop_scratch"][name] = n
spilled = {k: v for k, v in res["unit_loop_scratch"].items() if v}
print(f"# unit loops with scratch instructions: {len(spilled)} of {len(res['unit_loop_scratch'])} kernels with marked phases")
for k, v in sorted(spilled.items()):
    print(f"#   {v:3d}  " + subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:150])
json.dump(res, open(os.path.join(ROOT, "profiles", "isa_hist.json"), "w"), indent=1)
