"""Mechanical comparison of the three statements of the C ABI: include/fugue_amd.h (the contract), rust/fugue-gpu/src/ffi.rs (the
reference-side binding, text that has never met a compiler) and the ctypes declarations of fugue_amd/engine.py.  No compiler needed:
prototypes, `#[repr(C)]` structs and callback typedefs are parsed into canonical signatures -- argument count, pointer-ness and
const-ness, integer width, field order -- and compared.  Used by tests/test_boundary_cpu.py."""
from __future__ import annotations

import ctypes
import re

C_SCALARS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "unsigned": "u32", "int64_t": "i64", "long long": "i64",
             "uint64_t": "u64", "unsigned long long": "u64", "double": "f64", "size_t": "usize", "char": "char", "void": "void"}
RUST_SCALARS = {"c_int": "i32", "i32": "i32", "u32": "u32", "i64": "i64", "u64": "u64", "f64": "f64", "usize": "usize",
                "c_char": "char", "c_void": "void", "u8": "u8"}
CALLBACKS = ("fg_acov_fn", "fg_reduce_fn")


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_type(t):
    """'const double *' -> ('ptr', 'const', 'f64');  'int' -> 'i32';  'fg_engine *' -> ('ptr', 'mut', 'fg_engine')"""
    t = " ".join(t.replace("*", " * ").split())
    depth = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = " ".join(w for w in base if w not in ("const", "struct"))
    canon = C_SCALARS.get(base, base)
    for level in range(depth):
        canon = ("ptr", "const" if (const and level == 0) else "mut", canon)
    return canon


def _c_decl(decl):
    """one parameter / field declaration without its name -> canonical type"""
    decl = decl.strip()
    if decl == "void":
        return None
    m = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", decl)
    if m and m.group(1).strip() and m.group(1).strip() not in ("const", "unsigned", "long", "unsigned long"):
        decl = m.group(1)
    return _c_type(decl)


def parse_header(text):
    text = _strip_c(text)
    out = {"fns": {}, "structs": {}, "callbacks": {}}
    for name, body in re.findall(r"typedef struct (\w+)\s*\{([^}]*)\}", text):
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first, *more = [s.strip() for s in stmt.split(",")]
            m = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", first)
            ty = _c_type(m.group(1))
            fields.append((m.group(2), ty))
            fields += [(n, ty) for n in more]
        out["structs"][name] = fields
    for ret, name, args in re.findall(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\(([^)]*)\)\s*;", text):
        out["callbacks"][name] = (_c_type(ret), [_c_decl(a) for a in args.split(",") if _c_decl(a) is not None])
    text = re.sub(r"typedef[^;{]*\([^;]*;", " ", text)
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(fg_[a-z0-9_]+)\s*\(([^;{}()]*)\)\s*;", text):
        if ret.strip().startswith("typedef"):
            continue
        params = [_c_decl(a) for a in args.split(",")] if args.strip() else []
        out["fns"][name] = (_c_type(ret), [p for p in params if p is not None])
    return out


def _rust_type(t):
    t = t.strip()
    if t.startswith("*const "):
        return ("ptr", "const", _rust_type(t[7:]))
    if t.startswith("*mut "):
        return ("ptr", "mut", _rust_type(t[5:]))
    return RUST_SCALARS.get(t, t)


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
    return parts


def parse_rust(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {"fns": {}, "structs": {}, "callbacks": {}}
    for name, body in re.findall(r"#\[repr\(C\)\][^{;]*?pub struct (\w+)\s*\{([^}]*)\}", text):
        fields = []
        for f in _split_top(body):
            m = re.match(r"\s*(?:pub\s+)?(\w+)\s*:\s*(.+?)\s*$", f, flags=re.S)
            if m:
                fields.append((m.group(1), _rust_type(m.group(2))))
        out["structs"][name] = fields
    for name, args, ret in re.findall(r"pub type (\w+)\s*=\s*Option<\s*unsafe extern \"C\" fn\(([^)]*)\)\s*(?:->\s*([\w\s\*]+?))?\s*>\s*;", text):
        out["callbacks"][name] = (_rust_type(ret) if ret else "void", [_rust_type(a.split(":", 1)[1]) for a in _split_top(args)])
    for name, args, ret in re.findall(r"pub fn (fg_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", text):
        out["fns"][name] = (_rust_type(ret) if ret else "void", [_rust_type(a.split(":", 1)[1]) for a in _split_top(args)])
    return out


def compare_header_rust(header_text, rust_text):
    """-> list of human-readable mismatches (empty: the binding states the header's ABI)"""
    H, R = parse_header(header_text), parse_rust(rust_text)
    bad = []
    for name, (ret, params) in sorted(H["fns"].items()):
        if name not in R["fns"]:
            bad.append(f"{name}: not bound")
            continue
        rret, rparams = R["fns"][name]
        if rret != ret:
            bad.append(f"{name}: returns {rret}, header {ret}")
        if len(rparams) != len(params):
            bad.append(f"{name}: {len(rparams)} arguments, header {len(params)}")
            continue
        for i, (a, b) in enumerate(zip(rparams, params)):
            if a != b:
                bad.append(f"{name}: argument {i} is {a}, header {b}")
    for name in sorted(set(R["fns"]) - set(H["fns"])):
        bad.append(f"{name}: bound but not in the header")
    for name, fields in sorted(H["structs"].items()):
        if name not in R["structs"]:
            bad.append(f"struct {name}: no #[repr(C)] struct")
            continue
        if R["structs"][name] != fields:
            bad.append(f"struct {name}: fields {R['structs'][name]}, header {fields}")
    for name, sig in sorted(H["callbacks"].items()):
        if R["callbacks"].get(name) != sig:
            bad.append(f"callback {name}: {R['callbacks'].get(name)}, header {sig}")
    return bad


# ---- ctypes side
_CT_SCALARS = {ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_uint32: "u32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64",
               ctypes.c_double: "f64", ctypes.c_size_t: "usize", ctypes.c_longlong: "i64", ctypes.c_ulonglong: "u64"}


def _ctypes_matches(ct, canon, structs):
    """does the ctypes declaration `ct` pass an argument the way the header's `canon` type expects?"""
    if isinstance(canon, tuple):                                  # a pointer
        if ct in (ctypes.c_void_p, ctypes.c_char_p):
            return ct is ctypes.c_void_p or canon[2] == "char"
        if isinstance(ct, type) and issubclass(ct, ctypes._Pointer):
            tgt = ct._type_
            if tgt is ctypes.c_void_p:
                return isinstance(canon[2], tuple)                # void **
            if tgt in _CT_SCALARS:
                return _CT_SCALARS[tgt] == canon[2]
            if isinstance(tgt, type) and issubclass(tgt, ctypes.Structure):
                return tgt.__name__ == canon[2] and structs.get(canon[2]) == [(n, _CT_SCALARS.get(t, t)) for n, t in tgt._fields_]
            return False
        return isinstance(ct, type) and issubclass(ct, ctypes._CFuncPtr)
    if canon in CALLBACKS:
        return (isinstance(ct, type) and issubclass(ct, ctypes._CFuncPtr)) or ct is ctypes.c_void_p
    return _CT_SCALARS.get(ct) == canon or (_CT_SCALARS.get(ct) in ("u64", "usize") and canon in ("u64", "usize"))


def compare_header_ctypes(header_text, lib):
    H = parse_header(header_text)
    bad = []
    for name, (ret, params) in sorted(H["fns"].items()):
        fn = getattr(lib, name)
        argtypes = list(fn.argtypes or [])
        if len(argtypes) != len(params):
            bad.append(f"{name}: ctypes declares {len(argtypes)} arguments, header {len(params)}")
            continue
        for i, (ct, canon) in enumerate(zip(argtypes, params)):
            if not _ctypes_matches(ct, canon, H["structs"]):
                bad.append(f"{name}: ctypes argument {i} is {ct}, header {canon}")
        rt = fn.restype
        if ret == "void":
            ok = rt is None or rt is ctypes.c_int                 # (a void result read as the default int is never used)
        elif isinstance(ret, tuple):
            ok = rt in (ctypes.c_void_p, ctypes.c_char_p)
        else:
            ok = _CT_SCALARS.get(rt) == ret
        if not ok:
            bad.append(f"{name}: ctypes restype {rt}, header {ret}")
    return bad
