#!/usr/bin/env python3
"""Generate rust-shim/plonk-gpu-sys/src/lib.rs from include/ark_plonk_amd.h.

The reference is Rust (heliaxdev/ark-plonk); this image has no cargo/rustc, so the shim cannot be compiled here.  What CAN be
checked here is that the Rust `extern "C"` block says exactly what the C header says: this script derives one from the other,
`tests/test_rust_shim.py` re-runs it (the committed file must be what it prints) and re-parses both sides independently.

    python tools/gen_rust_ffi.py            # writes rust-shim/plonk-gpu-sys/src/lib.rs
    python tools/gen_rust_ffi.py --stdout   # prints it
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ark_plonk_amd.h")
OUT = os.path.join(ROOT, "rust-shim", "plonk-gpu-sys", "src", "lib.rs")

# C scalar -> Rust
SCALAR = {
    "int": "i32", "int64_t": "i64", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64",
    "char": "c_char", "void": "c_void",
}
OPAQUE = {"zk_ctx": "ZkCtx", "zk_srs": "ZkSrs", "zk_transcript": "ZkTranscript"}
STRUCTS = {"zk_domain_info": "ZkDomainInfo", "zk_quotient_args": "ZkQuotientArgs", "zk_proof": "ZkProof"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_type(ctype: str) -> str:
    """`const uint64_t* const*` -> `*const *const u64`; `zk_ctx**` -> `*mut *mut ZkCtx`; `const void* const[4]` -> `*const *const c_void`."""
    t = " ".join(ctype.replace("*", " * ").split())
    arr = re.search(r"\[(\d*)\]$", t)
    if arr:                                    # an array parameter decays to a pointer to its element type
        t = t[: arr.start()].strip() + " *"
    toks = t.split()
    # base type with an optional leading const
    const_base = False
    if toks and toks[0] == "const":
        const_base = True
        toks = toks[1:]
    base = toks[0]
    rest = toks[1:]
    if base in SCALAR:
        r = SCALAR[base]
    elif base in OPAQUE:
        r = OPAQUE[base]
    elif base in STRUCTS:
        r = STRUCTS[base]
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    # each '*' optionally followed by 'const' (constness of that pointer level, irrelevant for a by-value parameter except
    # as the pointee constness of the NEXT level)
    pointee_const = const_base
    i = 0
    while i < len(rest):
        if rest[i] != "*":
            raise ValueError(f"cannot parse {ctype!r}")
        r = ("*const " if pointee_const else "*mut ") + r
        pointee_const = False
        if i + 1 < len(rest) and rest[i + 1] == "const":
            pointee_const = True
            i += 1
        i += 1
    return r


RUST_KEYWORDS = {"in": "input", "type": "ty", "ref": "r", "mod": "modulus", "fn": "f", "box": "b", "move": "mv", "match": "m", "loop": "lp"}


def split_params(args: str):
    out = []
    for a in args.split(","):
        a = " ".join(a.split())
        if not a or a == "void":
            continue
        m = re.match(r"(.*?)(\w+)\s*(\[\w*\])?$", a)
        ctype = (m.group(1).strip() + (m.group(3) or "")).strip()
        out.append((m.group(2), ctype))
    return out


def parse_header(text: str):
    body = strip_comments(text)
    protos = []
    for m in re.finditer(r"\n((?:const\s+)?(?:int|void|size_t|char|zk_transcript)\s*\*?\s*)(zk_\w+)\s*\(([^;{]*)\)\s*;", body):
        ret = " ".join(m.group(1).split())
        protos.append((m.group(2), ret, split_params(m.group(3))))
    defines = re.findall(r"#define\s+(ZK_\w+)\s+\(?(-?\w+)\)?", body)
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        if m.group(3) not in STRUCTS:
            continue
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `const void *w_l, *w_r` / `uint64_t alpha[4], beta[4]` / `const void* sigma[4]` / `const char* const* custom_labels`
            if "," not in decl:
                (nm, ctype), = split_params(decl)
                arr = re.search(r"\[(\w+)\]$", ctype)
                n_arr = None
                if arr:
                    n_arr = arr.group(1)
                    ctype = ctype[: arr.start()].strip()
                fields.append((nm, ctype, n_arr))
                continue
            mm = re.match(r"((?:const\s+)?\w+)\s*(.*)$", decl)
            base, names = mm.group(1), mm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                stars = nm.count("*")
                nm = nm.replace("*", "").strip()
                arr = re.search(r"\[(\w+)\]$", nm)
                n_arr = None
                if arr:
                    n_arr = arr.group(1)
                    nm = nm[: arr.start()]
                fields.append((nm, base + "*" * stars, n_arr))
        structs.append((m.group(3), fields))
    return protos, defines, structs


def generate(text: str) -> str:
    protos, defines, structs = parse_header(text)
    o = []
    o.append("//! Raw FFI of `libark_plonk_amd.so` (the MI355X NTT + MSM hot path).  GENERATED by tools/gen_rust_ffi.py from")
    o.append("//! include/ark_plonk_amd.h -- do not edit; tests/test_rust_shim.py fails when the two drift apart.")
    o.append("#![allow(non_camel_case_types, clippy::too_many_arguments)]")
    o.append("use core::ffi::c_void;")
    o.append("use std::os::raw::c_char;")
    o.append("")
    for c_name, r_name in OPAQUE.items():
        o.append(f"/// opaque `{c_name}`")
        o.append("#[repr(C)]")
        o.append(f"pub struct {r_name} {{ _private: [u8; 0] }}")
    o.append("")
    const_vals = {}
    for name, val in defines:
        if name.endswith("_H"):
            continue
        if re.fullmatch(r"-?\d+", val):
            ty = "i32" if (name.startswith("ZK_ERR") or name in ("ZK_OK",) or name.startswith("ZK_CURVE")) else "u32"
            o.append(f"pub const {name}: {ty} = {val};")
            const_vals[name] = val
        elif re.fullmatch(r"0x[0-9a-fA-F]+u?", val):
            o.append(f"pub const {name}: u32 = {val.rstrip('u')};")
            const_vals[name] = str(int(val.rstrip("u"), 16))
    o.append("")
    for s_name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append(f"pub struct {STRUCTS[s_name]} {{")
        for nm, ctype, n_arr in fields:
            rt = rust_type(ctype)
            if n_arr is not None:
                n = const_vals.get(n_arr, n_arr)
                rt = f"[{rt}; {n}]"
            o.append(f"    pub {nm}: {rt},")
        o.append("}")
        o.append("")
    o.append('#[link(name = "ark_plonk_amd")]')
    o.append('extern "C" {')
    for name, ret, params in protos:
        ps = ", ".join(f"{RUST_KEYWORDS.get(pn, pn)}: {rust_type(pt)}" for pn, pt in params)
        rr = "" if ret == "void" else f" -> {rust_type(ret)}"
        o.append(f"    pub fn {name}({ps}){rr};")
    o.append("}")
    o.append("")
    return "\n".join(o)


def main():
    text = open(HEADER).read()
    src = generate(text)
    if "--stdout" in sys.argv:
        sys.stdout.write(src)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(src)
    print(OUT)


if __name__ == "__main__":
    main()
