#!/usr/bin/env python3
"""Diffs every #[repr(C)] struct of the Rust binding (patches/0001-hiphase-gpu.patch: src/gpu_ffi.rs) against the layout the
library was compiled with (hp_abi_layout()), without a Rust toolchain: the structs' text is parsed, sizes / alignments / field
offsets are computed by the C rules #[repr(C)] follows, and compared name by name. A struct is tied to its C counterpart by the
`/// C: <name>` line above it.   usage: check_rust_layout.py [gpu_ffi.rs] -> exit 0 / 1; importable: check(path) -> list of problems"""
import ctypes as C
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIM = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4),
        "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "usize": (8, 8), "isize": (8, 8)}


def gpu_ffi_from_patch(patch_path):
    """the text of src/gpu_ffi.rs as the patch creates it"""
    out, on = [], False
    for line in open(patch_path).read().split("\n"):
        if line.startswith("diff --git"):
            on = line.endswith("b/src/gpu_ffi.rs")
            continue
        if on and line.startswith("+") and not line.startswith("+++"):
            out.append(line[1:])
    return "\n".join(out)


def library_layout():
    for name in ("libhiphase_capture.so", "libhiphase_gpu.so"):
        p = os.path.join(ROOT, "hiphase_amd", name)
        if os.path.exists(p):
            try:
                dll = C.CDLL(p)
            except OSError:
                continue
            dll.hp_abi_layout.restype = C.c_char_p
            return json.loads(dll.hp_abi_layout().decode()), dll
    raise SystemExit("build hiphase_amd/libhiphase_gpu.so or `make -C hiphase_amd/csrc capture` first")


def parse(text):
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): usize = (\d+);", text)}
    structs = {}
    for m in re.finditer(r"/// C: (\w+)\n((?:#\[[^\]]*\]\n)*)pub struct (\w+) \{(.*?)\n\}", text, re.S):
        cname, attrs, rname, body = m.groups()
        assert "#[repr(C)]" in attrs, f"{rname} is not #[repr(C)]"
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^,\n]+),?", body)]
        structs[rname] = (cname, fields)
    return consts, structs


def layout_of(ty, consts, structs, memo):
    ty = ty.strip()
    if ty.startswith("*const ") or ty.startswith("*mut "):
        return 8, 8
    if ty in PRIM:
        return PRIM[ty]
    m = re.match(r"\[(.+); (\w+)\]$", ty)
    if m:
        size, align = layout_of(m.group(1), consts, structs, memo)
        n = int(m.group(2)) if m.group(2).isdigit() else consts[m.group(2)]
        return size * n, align
    if ty in structs:
        return struct_layout(ty, consts, structs, memo)[:2]
    raise ValueError(f"type {ty!r}")


def struct_layout(rname, consts, structs, memo):
    if rname in memo:
        return memo[rname]
    off, align, offsets = 0, 1, {}
    for fname, ty in structs[rname][1]:
        s, a = layout_of(ty, consts, structs, memo)
        off = (off + a - 1) // a * a
        offsets[fname] = off
        off += s
        align = max(align, a)
    memo[rname] = ((off + align - 1) // align * align, align, offsets)
    return memo[rname]


def check(text):
    lay, dll = library_layout()
    consts, structs = parse(text)
    problems, memo = [], {}
    if not structs:
        problems.append("no `/// C: name` + #[repr(C)] structs found")
    for rname, (cname, fields) in structs.items():
        if cname not in lay:
            problems.append(f"{rname}: the library has no struct {cname}")
            continue
        size, align, offsets = struct_layout(rname, consts, structs, memo)
        want = lay[cname]
        if size != want["sizeof"] or align != want["alignof"]:
            problems.append(f"{rname} / {cname}: sizeof {size} alignof {align}, library {want['sizeof']} / {want['alignof']}")
        if list(offsets) != list(want["fields"]):
            problems.append(f"{rname} / {cname}: fields {list(offsets)} != {list(want['fields'])}")
        for f, o in offsets.items():
            if want["fields"].get(f) != o:
                problems.append(f"{rname}.{f}: offset {o}, library {want['fields'].get(f)}")
    # the names check_layout() passes to hp_abi_sizeof / hp_abi_offsetof at start-up must be ones the library knows
    dll.hp_abi_sizeof.restype = C.c_size_t
    dll.hp_abi_offsetof.restype = C.c_size_t
    for m in re.finditer(r'check_struct!\((\w+), "(\w+)", \[([^\]]*)\]\)', text, re.S):
        rname, cname, flist = m.groups()
        if rname not in structs or structs[rname][0] != cname:
            problems.append(f"check_struct!({rname}, {cname}): not the pair the struct's `/// C:` line names")
            continue
        listed = [f.strip() for f in flist.replace("\n", " ").split(",") if f.strip()]
        if listed != [f for f, _ in structs[rname][1]]:
            problems.append(f"check_struct!({rname}): checks {listed}, the struct has {[f for f, _ in structs[rname][1]]}")
        if dll.hp_abi_sizeof(cname.encode()) != struct_layout(rname, consts, structs, memo)[0]:
            problems.append(f"hp_abi_sizeof({cname}) != {rname}")
        for f in listed:
            if dll.hp_abi_offsetof(cname.encode(), f.encode()) != memo[rname][2].get(f):
                problems.append(f"hp_abi_offsetof({cname}, {f}) != {rname}.{f}")
    checked = {m.group(1) for m in re.finditer(r"check_struct!\((\w+),", text)}
    for rname in structs:
        if rname not in checked:
            problems.append(f"{rname} is not covered by check_layout()")
    return problems, sorted(structs)


if __name__ == "__main__":
    srcs = sys.argv[1:] or [os.path.join(ROOT, "patches", n) for n in sorted(os.listdir(os.path.join(ROOT, "patches"))) if n.endswith(".patch")]
    text = "\n".join(gpu_ffi_from_patch(src) if src.endswith(".patch") else open(src).read() for src in srcs)
    problems, names = check(text)
    for p in problems:
        print("MISMATCH:", p)
    print(f"{len(names)} #[repr(C)] structs checked against the library: {', '.join(names)}" if not problems else f"{len(problems)} problems")
    sys.exit(1 if problems else 0)
