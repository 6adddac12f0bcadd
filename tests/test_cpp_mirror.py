"""The C++ host mirror above the C ABI (include/hiphase_gpu.hpp) and its C++ parity tests (tests/cpp/mirror_test.cpp).

CPU: the header compiles on its own, the test program builds and links, and without a GPU it fails loudly.
GPU: the built-in tests (the reference's known answers + GPU == oracle through the mirror) pass, and `solve_block`
through the C++ mirror prints exactly what the Python mirror computes for the same decoded block — the Python side is
held to the oracle-assembled pipeline by tests/test_e2e_gpu.py and tests/test_local_gpu.py on the same generators."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "mirror_test")


def build_binary():
    import __graft_entry__ as g
    g.build()          # libhiphase_gpu.so, liboracle.so and tests/cpp/mirror_test
    assert os.path.exists(BIN)


def test_header_is_self_contained_and_test_program_links():
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++",
                        os.path.join(ROOT, "include", "hiphase_gpu.hpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    build_binary()


def test_cpp_mirror_fails_loudly_without_a_gpu(hp_lib):
    if hp_lib.hp_device_count() > 0:
        pytest.skip("a GPU is visible")
    build_binary()
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


# ---- GPU ----------------------------------------------------------------------------------------------------------
def _tok(b):
    return bytes(b).decode() if len(b) else "-"


def write_case(path, reference, hets, homs, records, cfg, global_mode, min_matched=2):
    from hiphase_amd.read_parsing import CIGAR_OPS
    with open(path, "w") as f:
        f.write(f"reference 0 {_tok(reference)}\n")
        f.write(f"params {min_matched} 1000 3 {cfg.max_edit_distance} {cfg.wfa_prune_distance} {cfg.global_failure_ratio!r} "
                f"{cfg.global_failure_minimum} {1 if global_mode else 0}\n")
        for tag, vs in (("V", hets), ("H", homs)):
            for v in vs:
                f.write(f"{tag} {int(v.variant_type)} {v.position} {v.ref_len} {_tok(v.allele0)} {_tok(v.allele1)} "
                        f"{v.index_allele0} {v.index_allele1} {1 if v.is_ignored else 0} {_tok(v.prefix)} {_tok(v.postfix)}\n")
        for r in records:
            line = f"R {r.qname} {r.min_position} {r.max_position} {_tok(r.read_align)} {1 if r.local is not None else 0}"
            if r.local is not None:
                cg = [(int(n) << 4) | (CIGAR_OPS.index(op) if isinstance(op, str) else int(op)) for op, n in r.local.cigar]
                line += f" {r.local.pos} {len(cg)} " + " ".join(str(c) for c in cg)
                line += f" {_tok(r.local.seq)} {bytes(r.local.qual).hex() or '-'}"
            f.write(line + "\n")


def canonical(res, segs, stats):
    out = ["h1 " + "".join(str(int(x)) for x in res.haplotype_1), "h2 " + "".join(str(int(x)) for x in res.haplotype_2),
           "stats " + " ".join(str(int(x)) for x in res.statistics),
           "block_ids" + "".join(f" {t}" for t in res.block_ids),
           "sub_blocks" + "".join(" " + ",".join(str(i) for i in b) for b in res.sub_phase_blocks),
           f"load {stats.num_reads} {stats.skipped_reads} {stats.global_aligned} {stats.local_aligned}"]
    for s in segs:
        out.append(f"segment {s.read_name} {s.start} {s.end} {''.join(str(a) for a in s.alleles)} {bytes(s.quals).hex() or '-'}")
    out += [f"haplotag {k} {v[0]} {v[1]}" for k, v in res.haplotags.items()]
    return out


def run_case(path):
    r = subprocess.run([BIN, "--case", path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().split("\n")


@pytest.mark.gpu
def test_cpp_builtin_tests():
    build_binary()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 failed" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_cpp_solve_block_global_mode(tmp_path, seed):
    from e2e_util import make_block
    from hiphase_amd.phaser import solve_block
    from hiphase_amd.read_parsing import GlobalRealignmentConfig, load_full_read_segments
    build_binary()
    ref, hets, homs, records, _ = make_block(seed)
    cfg = GlobalRealignmentConfig()
    res, _, segs = solve_block(7, records, hets, homs, ref, global_config=cfg)
    _, _, stats = load_full_read_segments(records, hets, homs, ref, config=cfg)
    path = str(tmp_path / "case.txt")
    write_case(path, ref, hets, homs, records, cfg, True)
    assert run_case(path) == canonical(res, segs, stats)


@pytest.mark.gpu
@pytest.mark.parametrize("max_ed,minimum,ratio", [(4, 5, 0.3), (8, 10, 0.9)])
def test_cpp_solve_block_fallback_replay(tmp_path, max_ed, minimum, ratio):
    """WFA failures -> local re-alignment, with and without the `global_disabled` flip (read_parsing.rs:556-600)."""
    from local_util import make_local_block
    from test_local_gpu import to_aligned
    from hiphase_amd.phaser import solve_block
    from hiphase_amd.read_parsing import GlobalRealignmentConfig, load_full_read_segments
    build_binary()
    ref, variants, _, lrecs = make_local_block(21, ref_len=20000, n_vars=100, n_reads=120, read_len=(800, 3000), noise=0.004)
    hets = [v for v in variants if int(v.variant_type) in (0, 1, 2, 3)]
    records = [to_aligned(r) for r in lrecs if any(op in "M=X" for op, _ in r.cigar)]
    cfg = GlobalRealignmentConfig(max_edit_distance=max_ed, wfa_prune_distance=max_ed, global_failure_minimum=minimum,
                                  global_failure_ratio=ratio)
    res, _, segs = solve_block(7, records, hets, [], ref, global_config=cfg)
    _, _, stats = load_full_read_segments(records, hets, [], ref, config=cfg)
    assert stats.local_aligned > 0
    path = str(tmp_path / "case.txt")
    write_case(path, ref, hets, [], records, cfg, True)
    assert run_case(path) == canonical(res, segs, stats)


@pytest.mark.gpu
def test_cpp_solve_block_local_mode(tmp_path):
    """--disable-global-realignment (phaser.rs:521-537): every record through hp_local_realign_batch."""
    from local_util import make_local_block
    from test_local_gpu import to_aligned
    from hiphase_amd.phaser import solve_block
    from hiphase_amd.read_parsing import AlignedRecord, GlobalRealignmentConfig, load_read_segments
    build_binary()
    ref, variants, _, lrecs = make_local_block(11, ref_len=20000, n_vars=120, n_reads=200, read_len=(1500, 6000))
    res, _, segs = solve_block(7, lrecs, variants, [], ref, global_realignment=False)
    _, _, stats, _ = load_read_segments(lrecs, variants)
    records = [to_aligned(r) if any(op in "M=X" for op, _ in r.cigar) else AlignedRecord(r.qname, 0, 0, b"", r) for r in lrecs]
    path = str(tmp_path / "case.txt")
    write_case(path, ref, variants, [], records, GlobalRealignmentConfig(), False)
    assert run_case(path) == canonical(res, segs, stats)
