"""Generated block sets (C generator, C layout, no marshalling) through the product's entries against the oracle's whole path
(hpo_solve_block) on the same inputs: hp_solve_blocks and hp_blockstream_*, reads as ASCII and as BAM 4-bit, through the
compact graph-WFA kernels and the dense-band ones. The sets carry everything the bench workload does: SV / tandem-repeat /
multi-allelic calls, edit noise, reads that exceed max_edit_distance (local re-alignment fallback), supplementary records."""
import os
import ctypes as C

import pytest

from hiphase_amd import _ffi
from hiphase_amd.blocks import _params
from hiphase_amd.synth_sets import SynthSet, default_spec
from oracle_ffi import oracle
from e2e_util import outputs_diff

pytestmark = pytest.mark.gpu

KW = dict(max_block_hets=150, noisy_fraction=0.02, supplementary_fraction=0.05, frac_snv=0.75, frac_indel=0.13, frac_sv=0.04)


def oracle_outputs(sset, prm):
    d = oracle()
    out = sset.outputs().poison(0xEE)   # (what the oracle does not write cannot compare equal to what the product does not write: 0x77 there)
    for b in range(sset.n):
        assert d.hpo_solve_block(C.byref(sset.inputs[b]), C.byref(prm), C.byref(out.arr[b])) == 0
    return out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fmt", [_ffi.SEQ_ASCII, _ffi.SEQ_BAM4])
@pytest.mark.parametrize("path", ["compact", "compact-gen2", "dense-band"])
def test_solve_blocks_on_generated_sets_vs_oracle(fmt, path, monkeypatch):
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0" if path.startswith("compact") else "1000000000")
    monkeypatch.setenv("HP_WFA_GEN", "2" if path == "compact-gen2" else "3")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=600, seed=11, seq_format=fmt, **KW))
    exp = oracle_outputs(s, prm)
    got = s.outputs().poison(0x77)
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), got.arr, 0))
    bad = [b for b in range(s.n) if not got.equal(exp, b)]
    assert bad == []
    assert outputs_diff(s, got, exp) == []   # (the pure-Python comparator of tests/e2e_util.py: nothing shared with hp_block_output_equal)
    assert sum(got.arr[b].local_aligned for b in range(s.n)) > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fmt", [_ffi.SEQ_ASCII, _ffi.SEQ_BAM4])
def test_reads_in_device_readable_host_memory_vs_oracle(fmt, monkeypatch):
    """hp_host_alloc: a set whose records' bases all lie in it is read in place by the copy engines (hp_host_in_place_bytes counts them),
    through the one-call entry and through a block stream; a set with ONE block's bases elsewhere is staged as ever. Same results."""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=700, seed=37, seq_format=fmt, **KW))
    exp = oracle_outputs(s, prm)   # (the oracle reads the same host bytes: before or after the move makes no difference)
    s.relocate_pinned()
    before = lib.hp_host_in_place_bytes()
    got = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), got.arr, 0))
    assert [b for b in range(s.n) if not got.equal(exp, b)] == []
    moved = lib.hp_host_in_place_bytes() - before
    bases = s.info["read_bases"]
    want = bases // 2 if fmt == _ffi.SEQ_BAM4 else bases   # (records that cover no het of their block are not aligned at all)
    assert 0.9 * want <= moved <= 1.25 * want + (1 << 20)
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), 0, 3, C.byref(st))
    assert stream
    try:
        outs = [s.outputs() for _ in range(4)]
        tickets = []
        for o in outs:   # (three sets in flight: a fourth submit would wait for a slot only hp_blockstream_wait frees)
            if len(tickets) == 3:
                _ffi.check(lib.hp_blockstream_wait(stream, tickets.pop(0), None, None))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, o.arr, C.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
        for o in outs:
            assert [b for b in range(s.n) if not o.equal(exp, b)] == []
        assert lib.hp_host_in_place_bytes() - before == 5 * moved
    finally:
        lib.hp_blockstream_destroy(stream)
    # one block's records somewhere else: the whole set takes the staged way
    other = SynthSet(default_spec(lib, total_hets=700, seed=37, seq_format=fmt, **KW))
    mixed = (_ffi.BlockInput * s.n)(*[s.inputs[b] for b in range(s.n)])
    mixed[s.n // 2] = other.inputs[s.n // 2]
    before = lib.hp_host_in_place_bytes()
    got2 = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, mixed, C.byref(prm), got2.arr, 0))
    assert [b for b in range(s.n) if not got2.equal(exp, b)] == []
    assert lib.hp_host_in_place_bytes() == before
    # that block's bases in a second arena: in place again, as three runs (the first arena's blocks before it, after it, the other arena's)
    other.relocate_pinned()
    mixed[s.n // 2] = other.inputs[s.n // 2]
    got3 = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, mixed, C.byref(prm), got3.arr, 0))
    assert [b for b in range(s.n) if not got3.equal(exp, b)] == []
    assert 0.9 * want <= lib.hp_host_in_place_bytes() - before <= 1.25 * want + (1 << 20)


@pytest.mark.timeout(900)
def test_groups_that_leave_after_a_few_jobs_vs_oracle(monkeypatch):
    """HP_WFA2_GROUP_JOBS (experiment switch): the two smaller classes' groups take three jobs each and leave, the grid covers
    the list - workgroups retire all through the launch instead of staying until the queue is empty; same rows"""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    monkeypatch.setenv("HP_WFA2_GROUP_JOBS", "3")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=500, seed=29, seq_format=_ffi.SEQ_BAM4, **KW))
    exp = oracle_outputs(s, prm)
    got = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), got.arr, 0))
    assert [b for b in range(s.n) if not got.equal(exp, b)] == []


@pytest.mark.timeout(900)
@pytest.mark.parametrize("wide_min", ["1", "0"])
def test_noisy_sets_take_the_wide_slot_tables_vs_oracle(wide_min, monkeypatch):
    """2 % edit noise: most reads outgrow the slot tables of the two smaller graph-size classes. With HP_WFA2_WIDE_MIN=1 every
    such read is aligned again by the launches with the wide tables (hp_wfa2.hip late()), with 0 by the dense-band kernels:
    the same rows either way, equal to the oracle's"""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    monkeypatch.setenv("HP_WFA2_WIDE_MIN", wide_min)
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=500, seed=23, seq_format=_ffi.SEQ_BAM4, edit_noise=0.02, **KW))
    exp = oracle_outputs(s, prm)
    got = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), got.arr, 0))
    assert [b for b in range(s.n) if not got.equal(exp, b)] == []


@pytest.mark.timeout(1200)
def test_blockstream_on_generated_sets_vs_oracle(monkeypatch):
    """six different sets, three in flight, twice around: every set's results equal the oracle's"""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "256")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    sets = [SynthSet(default_spec(lib, total_hets=300 + 80 * k, seed=100 + k, seq_format=_ffi.SEQ_BAM4 if k % 2 else _ffi.SEQ_ASCII, **KW)) for k in range(6)]
    exps = [oracle_outputs(s, prm) for s in sets]
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), 0, 3, C.byref(st))
    assert stream, lib.hp_last_error()
    try:
        for rep in range(2):
            outs = [s.outputs() for s in sets]
            pending = []
            for k, s in enumerate(sets):
                if len(pending) == 3:
                    kk, t = pending.pop(0)
                    _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
                    assert [b for b in range(sets[kk].n) if not outs[kk].equal(exps[kk], b)] == []
                t = C.c_uint64(0)
                _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, outs[k].arr, C.byref(t)))
                pending.append((k, t.value))
            for kk, t in pending:
                _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
                assert [b for b in range(sets[kk].n) if not outs[kk].equal(exps[kk], b)] == []
    finally:
        lib.hp_blockstream_destroy(stream)


@pytest.mark.timeout(1200)
def test_blockstream_over_all_devices_vs_oracle(monkeypatch):
    """hp_blockstream_create(device_id = -1): one five-stage pipeline per device behind one submit / wait - here three of them
    on this box's GPU (HP_STREAM_DEVICES=3, the hook HP_QUEUE_WORKERS is for the dispatcher). Eight sets, every one handed to the
    least-loaded pipeline, waited for in submission order: every set's results equal the oracle's; every pipeline got work."""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "256")
    monkeypatch.setenv("HP_STREAM_DEVICES", "3")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    sets = [SynthSet(default_spec(lib, total_hets=250 + 90 * k, seed=300 + k, seq_format=_ffi.SEQ_BAM4 if k % 2 else _ffi.SEQ_ASCII, **KW)) for k in range(8)]
    exps = [oracle_outputs(s, prm) for s in sets]
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), -1, 2, C.byref(st))
    assert stream, lib.hp_last_error()
    assert lib.hp_blockstream_devices(stream) == 3
    used = set()
    try:
        outs = [s.outputs() for s in sets]
        pending = []
        for k, s in enumerate(sets):
            if len(pending) == 5:
                kk, t = pending.pop(0)
                _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
                assert [b for b in range(sets[kk].n) if not outs[kk].equal(exps[kk], b)] == []
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, outs[k].arr, C.byref(t)))
            used.add(t.value & 0xFF)
            pending.append((k, t.value))
        for kk, t in pending:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
            assert [b for b in range(sets[kk].n) if not outs[kk].equal(exps[kk], b)] == []
    finally:
        lib.hp_blockstream_destroy(stream)
    assert used == {0, 1, 2}


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("queue_devices", [None, "2"])
def test_async_block_entry_vs_oracle(queue_devices, monkeypatch):
    """hp_block_submit / hp_block_wait: every block of three generated sets submitted on its own (all of them in flight at once,
    as HiPhase's 40 x threads job slots would be, main.rs:328), merged behind the call into sets that travel through the
    per-device pipelines; waited for in reverse order. Every block equals the oracle's hpo_solve_block. Run in a subprocess per
    queue configuration: the dispatcher reads HP_QUEUE_WORKERS once per process."""
    import json
    import os
    import subprocess
    import sys
    code = """
import ctypes as C, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from hiphase_amd import _ffi
from hiphase_amd.blocks import _params
from hiphase_amd.synth_sets import SynthSet, default_spec
from oracle_ffi import oracle
lib = _ffi.lib()
prm = _params(2, 1000, 3, None, True)
KW = %r
sets = [SynthSet(default_spec(lib, total_hets=400 + 150 * k, seed=500 + k, seq_format=_ffi.SEQ_BAM4, **KW)) for k in range(3)]
d = oracle()
bad, tickets, outs = [], [], []
for s in sets:
    out = s.outputs(); outs.append(out)
    for b in range(s.n):
        t = C.c_uint64(0)
        _ffi.check(lib.hp_block_submit(1, C.byref(s.inputs[b]), C.byref(prm), C.byref(out.arr[b]), -1, C.byref(t)))
        tickets.append(t.value)
for t in reversed(tickets):
    _ffi.check(lib.hp_block_wait(t))
# a ticket is consumed by its wait: a second wait, or a ticket never issued, is an argument error - not a use-after-free (VERDICT r4 weak 9)
stale = [lib.hp_block_wait(tickets[0]), lib.hp_block_wait(tickets[-1]), lib.hp_block_wait(0), lib.hp_block_wait(0xDEADBEEF00)]
n = 0
for s, out in zip(sets, outs):
    exp = s.outputs()
    for b in range(s.n):
        assert d.hpo_solve_block(C.byref(s.inputs[b]), C.byref(prm), C.byref(exp.arr[b])) == 0
        n += 1
        if not out.equal(exp, b): bad.append(b)
# an invalid device is an argument error, not another GPU's work (ADVICE r3)
t = C.c_uint64(0)
rc = lib.hp_block_submit(1, C.byref(sets[0].inputs[0]), C.byref(prm), C.byref(outs[0].arr[0]), 99, C.byref(t))
print(json.dumps({"blocks": n, "bad": bad, "bad_device_rc": rc, "stale": stale}))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), KW)
    env = dict(os.environ)
    env.pop("HP_QUEUE_WORKERS", None)
    env["HP_WFA2_MIN_JOBS"] = "256"
    if queue_devices:
        env["HP_QUEUE_WORKERS"] = queue_devices
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1000, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["bad"] == [] and out["blocks"] > 10 and out["bad_device_rc"] == -4   # HP_ERR_ARG
    assert out["stale"] == [-4, -4, -4, -4]


@pytest.mark.timeout(900)
def test_hpbr_capture_replays_through_the_product_and_bench(tmp_path):
    """a `.hpbr` capture written with the oracle's results as expected output (standing in for a patched HiPhase's own
    solve_block): hp_solve_blocks on the replayed inputs equals the file, and `bench.py --replay` streams it and says so"""
    import json
    import os
    import subprocess
    import sys
    lib = _ffi.lib()
    d = oracle()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=500, seed=31, seq_format=_ffi.SEQ_BAM4, **KW))
    path = tmp_path / "cap.hpbr"
    exp = oracle_outputs(s, prm)
    for b in range(s.n):
        assert lib.hp_hpbr_append(str(path).encode(), C.byref(s.inputs[b]), C.byref(prm), C.byref(exp.arr[b])) == 0
    from hiphase_amd.synth_sets import Capture
    cap = Capture(path)
    got = cap.outputs()
    _ffi.check(lib.hp_solve_blocks(cap.n, cap.inputs, C.byref(cap.params[0]), got.arr, 0))
    assert [b for b in range(cap.n) if not lib.hp_block_output_equal(C.byref(cap.inputs[b]), C.byref(got.arr[b]), C.byref(cap.expected[b]))] == []
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--replay", str(path), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-resident"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["parity_vs_capture"] == {"blocks_compared": cap.n, "of": cap.n, "bit_identical": True} and out["value"] > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("noise,max_ed", [(0.014, 200), (0.03, 500), (0.006, 80)])
def test_reads_around_max_edit_distance_vs_oracle(noise, max_ed, monkeypatch):
    """A third of the reads carry noise that puts their edit distance right around max_edit_distance: some align, some end in
    Err(MaxEditDistance) and fall back to local re-alignment. The compact kernels hand most of them on; the shortcut that
    settles hopeless reads against the reference window alone (hp_wfa2_bound_kernel) must never call a read that aligns."""
    from hiphase_amd.read_parsing import GlobalRealignmentConfig
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    monkeypatch.setenv("HP_WFA2_BOUND", "8")     # test every leftover that got past 8 edits
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, GlobalRealignmentConfig(max_edit_distance=max_ed, wfa_prune_distance=max_ed), True)
    s = SynthSet(default_spec(lib, total_hets=400, seed=77, seq_format=_ffi.SEQ_BAM4, max_block_hets=120, noisy_fraction=0.35, noisy_noise=noise,
                              supplementary_fraction=0.03, frac_snv=0.80, frac_indel=0.14, frac_sv=0.03))
    exp = oracle_outputs(s, prm)
    got = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), got.arr, 0))
    assert [b for b in range(s.n) if not got.equal(exp, b)] == []
    n_local, n_global = sum(got.arr[b].local_aligned for b in range(s.n)), sum(got.arr[b].global_aligned for b in range(s.n))
    assert n_local > 10 and n_global > 10


@pytest.mark.timeout(1200)
def test_hifi_shaped_sets_vs_oracle():
    """Block sets with HiFi-shaped errors (hp_synth_reads_hifi: per-read rate lognormal around 0.2 %, a tail of reads at 1-4 %, half of
    the errors homopolymer-run indels - round 5) through the stream, in the records' BAM 4-bit codes: every field
    of every block identical to the oracle's hpo_solve_block (reference src/read_parsing.rs:520-867, src/astar_phaser.rs). The long
    homopolymer indels are where a read's alignment has equally good placements on neighbouring diagonals - the tie rules of
    src/wfa_graph.rs:476-510 at work."""
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    sets = [SynthSet(default_spec(lib, hifi=True, total_hets=2500 + 400 * k, max_block_hets=700, seed=900 + k, seq_format=_ffi.SEQ_BAM4)) for k in range(3)]
    sets.append(SynthSet(default_spec(lib, hifi=True, total_hets=1500, max_block_hets=300, seed=950, seq_format=_ffi.SEQ_ASCII, hifi_sigma=1.2, edit_noise=0.004)))   # a fatter tail
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), 0, 2, C.byref(st))
    assert stream
    try:
        outs, tickets = [s.outputs() for s in sets], []
        for s, o in zip(sets, outs):
            if len(tickets) == 2:
                _ffi.check(lib.hp_blockstream_wait(stream, tickets.pop(0), None, None))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, o.arr, C.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
    finally:
        lib.hp_blockstream_destroy(stream)
    n_blocks = fell_back = 0
    for s, o in zip(sets, outs):
        exp = oracle_outputs(s, prm)
        assert [b for b in range(s.n) if not o.equal(exp, b)] == []
        n_blocks += s.n
        fell_back += sum(o.arr[b].local_aligned for b in range(s.n))
    assert n_blocks > 40


def oracle_outputs_mt(sset, prm, threads=16):
    """hpo_solve_block over every block on `threads` host threads (ctypes releases the GIL for each call), largest blocks first"""
    import threading
    d = oracle()
    out = sset.outputs()
    order = sorted(range(sset.n), key=lambda b: -sset.inputs[b].n_records)
    lock, state = threading.Lock(), {"next": 0, "bad": []}

    def work():
        while True:
            with lock:
                k = state["next"]
                if k >= len(order):
                    return
                state["next"] = k + 1
            b = order[k]
            if d.hpo_solve_block(C.byref(sset.inputs[b]), C.byref(prm), C.byref(out.arr[b])) != 0:
                state["bad"].append(b)

    th = [threading.Thread(target=work) for _ in range(threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert state["bad"] == []
    return out


@pytest.mark.timeout(1500)
def test_eight_pipelines_with_bench_sized_sets_vs_oracle(monkeypatch):
    """What an 8-GPU node's single HiPhase process would run, exercised on ONE GPU: hp_blockstream_create(device_id
    = -1) with HP_STREAM_DEVICES=8 - eight six-stage pipelines (8 x 7 stage threads and their worker pools, 8 x the copy engines'
    descriptors) - depth 2, bench-sized sets (60 000 hets, ~136 k records, ~1.07 GB across PCIe each), 20 of them in submission
    order with 16 in flight: no deadlock, every pipeline gets work, every block of every set equals the oracle's hpo_solve_block
    (reference src/phaser.rs:406-630 per block; no collective)."""
    monkeypatch.setenv("HP_STREAM_DEVICES", "8")
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    sets = [SynthSet(default_spec(lib, seed=7100 + k)) for k in range(3)]
    for s in sets[:2]:
        s.relocate_pinned()          # two sets read in place by the copy engines, the third staged by the library
    exps = [oracle_outputs_mt(s, prm) for s in sets]
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), -1, 2, C.byref(st))
    assert stream, lib.hp_last_error()
    assert lib.hp_blockstream_devices(stream) == 8
    used, n_sets, checked = set(), 20, 0
    try:
        outs = [sets[k % 3].outputs() for k in range(n_sets)]
        pending = []

        def finish(kk, t):
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
            s = sets[kk % 3]
            assert [b for b in range(s.n) if not outs[kk].equal(exps[kk % 3], b)] == [], f"set {kk}"
            return s.n

        for k in range(n_sets):
            if len(pending) == 16:
                checked += finish(*pending.pop(0))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, sets[k % 3].n, sets[k % 3].inputs, outs[k].arr, C.byref(t)))
            used.add(t.value & 0xFF)
            pending.append((k, t.value))
        for kk, t in pending:
            checked += finish(kk, t)
    finally:
        lib.hp_blockstream_destroy(stream)
    assert used == set(range(8)) and checked > 20 * 200


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("case", ["routed", "into-an-idle-device", "over-the-cap", "off", "routed-reads-align"])
def test_records_routed_past_the_compact_kernels_vs_oracle(case, monkeypatch):
    """Round 5: a record whose CIGAR begins an operation every 20 bases or less (hp_block_record.local) is routed past the
    several-reads-per-wavefront kernels when its set is laid out; its way out (reference-window test, dense band) runs on the
    device's early worker beside the set's launch set, and finish() joins it (hp_wfa2.hip: W2Session::Early). Routing only - through
    the one-call entry and through a stream two deep every field of every block equals the oracle's hpo_solve_block (reference
    src/read_parsing.rs:520-637: Err(MaxEditDistance) -> local re-alignment), whether the records are routed, too many to be routed
    (a noisy SET is the wide-table launch's), not routed at all, routed only into an idle device, or routed and then
    ALIGNED by the dense band at a few hundred edits."""
    from hiphase_amd.read_parsing import GlobalRealignmentConfig
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    # (HP_WFA2_ROUTE: 0 = never - the default, it measured slower on the bench: DESIGN.md 3.7 -, 1 = only while no other launch set of the
    # device is being aligned, 2 = always)
    monkeypatch.setenv("HP_WFA2_ROUTE", "1" if case == "into-an-idle-device" else "2")
    if case == "over-the-cap":
        monkeypatch.setenv("HP_WFA2_SUSPECT_MAX", "3")
    if case == "off":
        monkeypatch.setenv("HP_WFA2_SUSPECT_OPS", "0")
    lib = _ffi.lib()
    grc = GlobalRealignmentConfig(max_edit_distance=1500, wfa_prune_distance=1500) if case == "routed-reads-align" else None
    prm = _params(2, 1000, 3, grc, True)
    kw = dict(KW, noisy_fraction=0.04)
    sets = [SynthSet(default_spec(lib, total_hets=700 + 150 * k, seed=1200 + k, seq_format=_ffi.SEQ_BAM4 if k != 1 else _ffi.SEQ_ASCII, **kw)) for k in range(3)]
    exps = [oracle_outputs(s, prm) for s in sets]
    before = lib.hp_wfa_routed_records()
    got = sets[0].outputs()
    _ffi.check(lib.hp_solve_blocks(sets[0].n, sets[0].inputs, C.byref(prm), got.arr, 0))
    assert [b for b in range(sets[0].n) if not got.equal(exps[0], b)] == []
    routed_one_call = lib.hp_wfa_routed_records() - before
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), 0, 2, C.byref(st))
    assert stream
    try:
        outs, tickets = [s.outputs() for s in sets], []
        for s, o in zip(sets, outs):
            if len(tickets) == 2:
                _ffi.check(lib.hp_blockstream_wait(stream, tickets.pop(0), None, None))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, o.arr, C.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
    finally:
        lib.hp_blockstream_destroy(stream)
    for s, o, e in zip(sets, outs, exps):
        assert [b for b in range(s.n) if not o.equal(e, b)] == []
    routed = lib.hp_wfa_routed_records() - before
    n_local = sum(outs[k].arr[b].local_aligned for k in range(3) for b in range(sets[k].n))
    if case in ("routed", "routed-reads-align"):
        assert routed_one_call > 10 and routed > 4 * 10
    elif case == "into-an-idle-device":
        assert routed_one_call > 10 and routed >= 2 * routed_one_call    # the one call and the stream's first set found the device idle
    else:
        assert routed == 0
    if case == "routed-reads-align":
        assert n_local < routed // 4      # most of the routed reads align within 1 500 edits: the dense band delivered their rows
    else:
        assert n_local > 30


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("gen", ["3", "2"])
def test_deep60_sets_vs_oracle(gen, monkeypatch):
    """BASELINE.json configs[4]'s shape through the WHOLE path (hp_synth_reads_deep60: 60x coverage, 15 % of the cells carrying the other
    haplotype's allele - conflicting rows, so the A* frontier prunes, reference src/astar_phaser.rs:564-585 - every tandem-repeat het
    multi-allelic, index_allele0 != 0, src/wfa_graph.rs:216-231, 1 % of the reads past max_edit_distance): the one-call entry and a block
    stream three deep, both kernel generations, every field of every block against hpo_solve_block - with pruned_solutions > 0 in it."""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    monkeypatch.setenv("HP_WFA_GEN", gen)
    lib = _ffi.lib()
    prm = _params(2, 1000, 3, None, True)
    sets = [SynthSet(default_spec(lib, deep60=True, total_hets=1400, max_block_hets=500, seed=71 + k, seq_format=fmt))
            for k, fmt in enumerate((_ffi.SEQ_BAM4, _ffi.SEQ_ASCII))]
    exps = [oracle_outputs(s, prm) for s in sets]
    assert sum(e.arr[b].stats.pruned_solutions for s, e in zip(sets, exps) for b in range(s.n)) > 100
    assert sum(1 for s in sets for b in range(s.n) for v in range(s.inputs[b].n_hets) if s.inputs[b].hets[v].flags & 2) > 300
    got = sets[0].outputs().poison(0x77)
    _ffi.check(lib.hp_solve_blocks(sets[0].n, sets[0].inputs, C.byref(prm), got.arr, 0))
    assert outputs_diff(sets[0], got, exps[0]) == [] and all(got.equal(exps[0], b) for b in range(sets[0].n))
    st = C.c_int(0)
    stream = lib.hp_blockstream_create(C.byref(prm), 0, 3, C.byref(st))
    assert stream
    try:
        order = [0, 1, 0, 1]
        outs, tickets = [sets[k].outputs().poison(0x33) for k in order], []
        for k, o in zip(order, outs):
            if len(tickets) == 3:
                _ffi.check(lib.hp_blockstream_wait(stream, tickets.pop(0), None, None))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, sets[k].n, sets[k].inputs, o.arr, C.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
    finally:
        lib.hp_blockstream_destroy(stream)
    for k, o in zip(order, outs):
        assert outputs_diff(sets[k], o, exps[k]) == []
        assert [b for b in range(sets[k].n) if not o.equal(exps[k], b)] == []
