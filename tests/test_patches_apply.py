"""patches/: the Rust side of the drop-in as unified diffs against HiPhase v1.5.0. No Rust
toolchain exists in this image, so what can be held is: the diffs apply to the reference tree (git apply --check, in order), every
#[repr(C)] struct of the binding has the layout the library was compiled with (scripts/check_rust_layout.py, which must also FAIL on a
perturbed struct), and every extern "C" function the binding declares is exported by the library with as many parameters as
include/hiphase_gpu.h gives it."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
REFERENCE = "/root/reference"
PATCHES = ["0001-hiphase-gpu.patch", "0002-hiphase-capture.patch", "0003-hiphase-gpu-async.patch"]


def _capture_lib():
    p = os.path.join(ROOT, "hiphase_amd", "libhiphase_capture.so")
    if not os.path.exists(p):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "hiphase_amd", "csrc"), "capture"], check=True)
    return p


def _rust_text():
    import check_rust_layout
    parts = [check_rust_layout.gpu_ffi_from_patch(os.path.join(ROOT, "patches", p)) for p in PATCHES]
    return check_rust_layout, "\n".join(parts)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")), reason="the reference tree is only present in the build container")
def test_patches_apply_in_order_to_the_reference_tree(tmp_path):
    tree = tmp_path / "hiphase"
    shutil.copytree(REFERENCE, tree, ignore=shutil.ignore_patterns(".git", "target"))
    for name in PATCHES:
        patch = os.path.join(ROOT, "patches", name)
        r = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=tree, capture_output=True, text=True)
        assert r.returncode == 0, f"{name}: {r.stderr}"
        r = subprocess.run(["git", "apply", "--whitespace=nowarn", patch], cwd=tree, capture_output=True, text=True)
        assert r.returncode == 0, f"{name}: {r.stderr}"
    ffi = (tree / "src" / "gpu_ffi.rs").read_text()
    phaser = (tree / "src" / "phaser.rs").read_text()
    main_rs = (tree / "src" / "main.rs").read_text()
    for text in (ffi, phaser, main_rs, (tree / "src" / "read_parsing.rs").read_text(), (tree / "build.rs").read_text()):
        code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', re.sub(r"//[^\n]*", "", text))   # (no string / comment contents)
        code = re.sub(r"'(?:[^'\\]|\\.)'", "' '", code)
        assert code.count("{") == code.count("}") and code.count("(") == code.count(")") and code.count("[") == code.count("]")
    # the call sites the patches name
    assert "read_parsing::gather_block_records(" in phaser and "crate::gpu_ffi::solve_block_gpu(&marshal)?" in phaser
    assert "crate::gpu_ffi::capture_block(" in phaser and "fn finish_block(" in phaser
    # the asynchronous form (0003): solve_block split into submit / finish, one finisher thread beside the pool (main.rs:326-462)
    assert "pub fn submit_block(" in phaser and "pub fn finish_block_gpu(" in phaser and "fn load_block_variants(" in phaser
    assert "hiphase::phaser::submit_block(" in main_rs and "hiphase::phaser::finish_block_gpu(submitted)" in main_rs
    assert main_rs.count('#[cfg(feature = "gpu")]') == 3 and main_rs.count('#[cfg(not(feature = "gpu"))]') == 1
    assert "pub struct PendingBlock" in ffi and "hp_block_submit(1, &p.marshal.input" in ffi and "hp_block_wait(self.ticket)" in ffi
    assert 'pub mod gpu_ffi;' in (tree / "src" / "lib.rs").read_text()
    cargo = (tree / "Cargo.toml").read_text()
    assert "gpu = []" in cargo and "capture = []" in cargo
    # the default build is untouched: everything new is behind a feature
    assert ffi.count('#[cfg(feature = "gpu")]') >= 4 and ffi.count('#[cfg(feature = "capture")]') >= 5
    # and the patched tree equals the one the second patch was cut from (no fuzz): applying in reverse gives the reference back
    for name in reversed(PATCHES):
        r = subprocess.run(["git", "apply", "-R", "--whitespace=nowarn", os.path.join(ROOT, "patches", name)], cwd=tree, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run(["diff", "-r", "-q", "-x", ".git", "-x", "target", REFERENCE, str(tree)], capture_output=True, text=True)
    assert r.stdout == "", r.stdout


def test_rust_structs_have_the_librarys_layout():
    _capture_lib()
    mod, text = _rust_text()
    problems, names = mod.check(text)
    assert problems == []
    assert {"HpBlockInput", "HpBlockOutput", "HpBlockParams", "HpBlockRecord", "HpLocalRead", "HpLocalVariant", "HpWfaVariant", "HpBlockView",
            "HpPhaseStats", "HpAstarParams"} <= set(names)


@pytest.mark.parametrize("edit", [
    ("    pub read_len: u32,\n    pub qname_id: u32,", "    pub qname_id: u32,\n    pub read_len: u32,"),        # two fields swapped
    ("    pub status: i32,\n    pub reserved: u32,\n    pub num_alleles", "    pub status: i32,\n    pub num_alleles"),   # a field dropped
    ("    pub read_offset: u32,\n    pub reserved: u32,\n}", "    pub read_offset: u64,\n    pub reserved: u32,\n}"),         # a wider type
    ("pub const HP_N_VARIANT_TYPES: usize = 11;", "pub const HP_N_VARIANT_TYPES: usize = 10;"),
])
def test_the_layout_check_is_falsifiable(edit):
    _capture_lib()
    mod, text = _rust_text()
    assert edit[0] in text
    problems, _ = mod.check(text.replace(edit[0], edit[1]))
    assert problems != []


def test_rust_externs_match_the_header():
    import ctypes as C
    _, text = _rust_text()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hiphase_gpu.h")).read(), flags=re.S)
    dll = C.CDLL(_capture_lib())
    externs = re.findall(r"pub fn (hp_\w+)\(([^)]*)\)", text)
    assert {n for n, _ in externs} >= {"hp_solve_blocks", "hp_block_submit", "hp_block_wait", "hp_last_error", "hp_abi_sizeof", "hp_abi_offsetof", "hp_hpbr_append", "hp_hpbk_append", "hp_hpbr_last_error"}
    host_only = {"hp_abi_sizeof", "hp_abi_offsetof", "hp_hpbr_append", "hp_hpbk_append", "hp_hpbr_last_error"}
    for name, args in externs:
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)\s*;", hdr)
        assert m, f"{name} is not declared in include/hiphase_gpu.h"
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        r_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), f"{name}: {len(r_args)} parameters in the binding, {len(c_args)} in the header"
        if name in host_only:
            assert hasattr(dll, name), f"libhiphase_capture.so lacks {name}"
