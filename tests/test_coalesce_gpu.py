"""Call coalescing behind the unchanged per-block entry points (hiphase_amd/csrc/hp_combine.h): HiPhase's worker pool
calls solve_block once per block from `--threads` threads (reference src/main.rs:385-408). 64 threads calling
hp_astar_solve / hp_solve_blocks at the same time must get bit-identical answers to the one-call-at-a-time path, much
faster, and nothing may hang when the threads exit."""
import threading
import time

import numpy as np
import pytest

from e2e_util import make_block
from hiphase_amd import _ffi, astar_solver, synth_block
from hiphase_amd.blocks import BlockSpec, solve_blocks

pytestmark = pytest.mark.gpu


def run_threads(n_threads, fn):
    out, errs = [None] * n_threads, []

    def body(t):
        try:
            out[t] = fn(t)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not any(x.is_alive() for x in th), "a caller never came back"
    assert not errs, errs
    return out, time.perf_counter() - t0


def test_concurrent_astar_solve_is_merged_and_identical():
    """Python threads: identical answers with the merging on and off (the rate is measured from C++ below: here the GIL
    and the marshalling dominate)."""
    n_thr, reps = 32, 4
    sizes = [15, 40, 90, 200, 25, 60, 12, 150]
    blocks = [[synth_block(sizes[(t + r) % len(sizes)], 30, 20, 0.02, 0.02, 1000 + 97 * t + r)[0] for r in range(reps)] for t in range(n_thr)]
    lib = _ffi.lib()

    def work(t):
        return [astar_solver(t, b) for b in blocks[t]]

    prev = lib.hp_set_coalescing(0)
    try:
        alone, _ = run_threads(n_thr, work)
        lib.hp_set_coalescing(1)
        merged, _ = run_threads(n_thr, work)
    finally:
        lib.hp_set_coalescing(prev)
    for a, m in zip(alone, merged):
        for x, y in zip(a, m):
            assert np.array_equal(x.haplotype_1, y.haplotype_1) and np.array_equal(x.haplotype_2, y.haplotype_2)
            assert x.statistics.as_tuple() == y.statistics.as_tuple()


@pytest.mark.timeout(900)
def test_worker_pool_rate_from_cpp():
    """64 std::threads x small blocks through hp_astar_solve (tests/cpp/coalesce_test.cpp), in a subprocess of THIS test (not of the
    collection): bit-identity with the one-launch-per-call answers is the hard assertion. The speed-up is a property of
    the machine's moment as well - the binary on its own reads 8-10 x (390-470 k against 45 k hets/s), beside a pytest process that
    holds a GPU context 4.4 x - so its bar is modest and tunable: HP_TEST_POOL_RATE (default 3), 0 switches it off."""
    import os
    import subprocess
    import __graft_entry__ as g
    g.build()
    binp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "coalesce_test")
    bar = os.environ.get("HP_TEST_POOL_RATE", "3")
    r = None
    for _attempt in range(2):
        r = subprocess.run([binp, "64", "12", bar], capture_output=True, text=True, timeout=400)
        print(r.stdout)
        assert "bit-identical" in r.stdout, r.stdout + r.stderr
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.timeout(900)
def test_worker_pool_with_recycled_device_blocks_poisoned():
    """The same binary with HP_DEV_CACHE_POISON=1 (every block the device-buffer cache hands out is filled with 0xA5 first) and no bar
    on the rate: merged batches mix blocks that take the segment-parallel heuristic (its tables uploaded on the segment stream) with
    blocks of the ordinary pass on the other stream. Round 6 had the ordinary pass look a block's segments up in those tables before
    they had arrived - harmless by luck on fresh memory, a GPU memory fault on poisoned memory; this run is what found it."""
    import os
    import subprocess
    import __graft_entry__ as g
    g.build()
    binp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "coalesce_test")
    env = dict(os.environ, HP_DEV_CACHE_POISON="1")
    r = subprocess.run([binp, "64", "12", "0"], capture_output=True, text=True, timeout=400, env=env)
    assert "bit-identical" in r.stdout and r.returncode == 0, r.stdout + r.stderr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("queue_devices", ["3", None])
def test_worker_pool_through_the_block_entry_over_all_devices(queue_devices):
    """64 std::threads each calling hp_solve_blocks(1, ..., device_id = -1) - what the one-call-site Rust patch does from HiPhase's
    worker pool - and then hp_block_submit / hp_block_wait with 40 tickets pending per thread (the asynchronous patch), with the
    dispatcher's queue served by three "devices" (HP_QUEUE_WORKERS=3 on this 1-GPU box: three pipelines) and by the visible ones:
    every block identical to one hp_solve_blocks call over all blocks (tests/cpp/dispatch_test.cpp); the rates are printed."""
    import json
    import os
    import subprocess
    import __graft_entry__ as g
    g.build()
    binp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "dispatch_test")
    env = dict(os.environ)
    env.pop("HP_QUEUE_WORKERS", None)
    if queue_devices:
        env["HP_QUEUE_WORKERS"] = queue_devices
    r = subprocess.run([binp, "64", "4000", "300", "2"], capture_output=True, text=True, timeout=800, env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    # blocking (T blocks in flight) and asynchronous (40 x T in flight) entries, every pass of both: identical to the one call
    assert out["mismatching_blocks"] == 0 and out["failed_calls"] == 0 and out["blocks"] > 64 and out["async_hets_per_s"] > 0


def test_concurrent_solve_blocks_is_merged_and_identical():
    n_thr = 16
    specs = []
    for t in range(n_thr):
        ref, hets, homs, records, _ = make_block(40 + t, ref_len=20000, n_hets=20 + t, n_homs=4, n_reads=40)
        specs.append(BlockSpec(t, ref, hets, homs, records))
    lib = _ffi.lib()
    prev = lib.hp_set_coalescing(0)
    try:
        alone, _ = run_threads(n_thr, lambda t: solve_blocks([specs[t]])[0])
        lib.hp_set_coalescing(1)
        merged, _ = run_threads(n_thr, lambda t: solve_blocks([specs[t]])[0])
    finally:
        lib.hp_set_coalescing(prev)
    for a, m in zip(alone, merged):
        assert np.array_equal(a.haplotype_1, m.haplotype_1) and np.array_equal(a.haplotype_2, m.haplotype_2)
        assert a.statistics == m.statistics and a.segments == m.segments and a.haplotags == m.haplotags
        assert a.span_counts.tolist() == m.span_counts.tolist()


def test_bad_block_in_a_merged_batch_only_fails_its_caller():
    good = [synth_block(30, 30, 20, 0.02, 0.02, 5 + t)[0] for t in range(7)]
    lib = _ffi.lib()
    prev = lib.hp_set_coalescing(1)
    try:
        def work(t):
            if t == 3:
                with pytest.raises(_ffi.HpError):
                    astar_solver(t, good[0], min_queue_size=10 ** 9)   # outside the packed-key limits: HP_ERR_UNSUPPORTED
                return None
            return astar_solver(t, good[t % len(good)])
        out, _ = run_threads(8, work)
    finally:
        lib.hp_set_coalescing(prev)
    assert sum(o is not None for o in out) == 7
