"""One case per process (tests/test_process_order_gpu.py runs it with subprocess): the ORDER in which a process first touches the
device decides how the HIP runtime waits for it (hp_runtime_wait_mode, hp_common.h ensure_runtime_flags). Prints one JSON line.

  lib-first       the library's first call is hp_wfa_assign_batch from worker threads (HiPhase's per-record form, reference
                  src/read_parsing.rs:769-780; nothing has asked for the device count), then a block stream is created, used
                  and destroyed - the sequence behind rounds 4-5's teardown hang
  framework-first torch initialises the device (allocation, kernel, synchronize) before the library is loaded; same work after
"""
import ctypes as C
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main(case):
    os.environ["HP_WFA2_MIN_JOBS"] = "0"
    if case == "framework-first":
        import torch
        x = torch.ones(1 << 20, device="cuda:0")
        y = (x * 2).sum().item()
        torch.cuda.synchronize()
        assert y == 2 << 20
    from hiphase_amd import _ffi
    from hiphase_amd.blocks import _params
    from hiphase_amd.synth_sets import SynthSet, default_spec
    from hiphase_amd.wfa_graph import wfa_assign_batch
    from wfa_util import synth_wfa_job
    from oracle_ffi import oracle
    lib = _ffi.lib()
    mode_before = lib.hp_runtime_wait_mode()
    specs = [synth_wfa_job(7100 + s, ref_len=1500 + 40 * s, n_vars=10, n_homs=3, noise=0.01)[0] for s in range(24)]
    res = [None] * 4
    whole = wfa_assign_batch(specs, prune_distance=500, max_edit_distance=500)   # (the main thread's own streams: they live as long as the process)
    os.environ["HP_WFA_GEN"] = "2"
    whole2 = wfa_assign_batch(specs, prune_distance=500, max_edit_distance=500)
    os.environ.pop("HP_WFA_GEN")

    def work(t):
        for _ in range(2):
            res[t] = wfa_assign_batch(specs[t::4], prune_distance=500, max_edit_distance=500)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    mode_after_generic = lib.hp_runtime_wait_mode()
    generic_ok = all(g[:3] == e[:3] and (g[3] == e[3]).all() for g, e in zip(whole2, whole)) and all(g[:3] == e[:3] and (g[3] == e[3]).all() for t in range(4) for g, e in zip(res[t], whole[t::4]))

    KW = dict(max_block_hets=150, noisy_fraction=0.02, supplementary_fraction=0.05, frac_snv=0.75, frac_indel=0.13, frac_sv=0.04)
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(lib, total_hets=700, seed=37, seq_format=_ffi.SEQ_ASCII, **KW))
    d = oracle()
    exp = s.outputs()
    for b in range(s.n):
        assert d.hpo_solve_block(C.byref(s.inputs[b]), C.byref(prm), C.byref(exp.arr[b])) == 0
    bad = []
    one = s.outputs()
    _ffi.check(lib.hp_solve_blocks(s.n, s.inputs, C.byref(prm), one.arr, 0))
    bad += [(-1, 0, b) for b in range(s.n) if not one.equal(exp, b)]
    for rep in range(2):   # (a stream created, used and destroyed, twice)
        st = C.c_int(0)
        stream = lib.hp_blockstream_create(C.byref(prm), 0, 3, C.byref(st))
        assert stream
        outs, tickets = [s.outputs() for _ in range(4)], []
        for o in outs:
            if len(tickets) == 3:
                _ffi.check(lib.hp_blockstream_wait(stream, tickets.pop(0), None, None))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, s.n, s.inputs, o.arr, C.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _ffi.check(lib.hp_blockstream_wait(stream, t, None, None))
        lib.hp_blockstream_destroy(stream)
        bad += [(rep, k, b) for k, o in enumerate(outs) for b in range(s.n) if not o.equal(exp, b)]
    freed = lib.hp_trim_device_cache()   # (hipFree synchronises every stream of the device: the other call that used to hang)
    print(json.dumps({"case": case, "mode_before": mode_before, "mode_after_generic": mode_after_generic, "mode": lib.hp_runtime_wait_mode(),
                      "generic_ok": bool(generic_ok), "blocks": s.n, "mismatches": len(bad), "trimmed": int(freed)}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
