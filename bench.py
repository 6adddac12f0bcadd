#!/usr/bin/env python3
"""bench.py — het variants phased / sec (BASELINE.json metric) through the WHOLE hot path on one MI355X per rank.

Default workload "path": synthetic read-bearing WGS-like block sets (hiphase_amd/csrc/hp_synth_reads.cpp: heavy-tailed block
sizes up to 4 165 hets, HiFi-like 15-kb reads at 30x with edit noise, a noisy tail that falls back to local re-alignment,
supplementary records, SNV / indel / SV / tandem-repeat calls, het + hom), STREAMED: one "step" = one NEW block set through
hp_blockstream_* (layout + PCIe of one set, graph-WFA of another and rows / A* / post-processing of a third overlap), its
bytes crossing PCIe inside the timed region; `value` = hets of all timed steps / wall time. `resident` is the secondary
inputs-already-in-HBM figure. Per-kernel rooflines in `kernels`, the CPU restatement on one and on all host cores beside it.

`--workload c2` is the solver-only figure of round 1 (BASELINE.json configs[1] shape: resident read x variant
matrices, N=5000, C=30, S=20), `--workload wgs` the solver on a heavy-tailed block-size mix.

Prints ONE JSON line (rank 0). See DESIGN.md §Measurement for the roofline accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 1.75  # l x (2-bit allele + 8-bit qual) + 2 x l x 2-bit haplotype


def make_blocks(args, rank):
    from hiphase_amd import synth_block
    from hiphase_amd.shard import rank_seeds
    blocks = []
    if args.replay:   # captured real blocks (.hpbk, hiphase_amd/block_io.py): every rank replays its LPT shard
        from hiphase_amd.block_io import read_blocks
        from hiphase_amd.shard import shard_lpt
        with open(args.replay, "rb") as f:
            allb = list(read_blocks(f))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        mine = shard_lpt([b.n_cells for b, _, _ in allb], world)[rank]
        args.replay_expected = [allb[i][2] for i in mine]
        return [allb[i][0] for i in mine]
    if args.workload == "c2":
        for seed in rank_seeds(20250509, rank, args.blocks):   # disjoint seed ranges per rank (weak scaling)
            blocks.append(synth_block(args.hets, args.coverage, args.span, args.error, 0.02, seed)[0])
    else:  # "wgs": heavy-tailed block sizes (docs/user_guide.md:257: median 15, mean ~220, max ~4000)
        import numpy as np
        rng = np.random.default_rng(12345 + rank)
        sizes = np.clip(np.exp(rng.normal(np.log(15.0), 2.2, args.blocks)).astype(int), 2, 4000)
        for i, n in enumerate(sizes):
            blocks.append(synth_block(int(n), args.coverage, args.span, args.error, 0.02, 777 + rank * 1000003 + i)[0])
    return blocks


def cpu_baseline(args, blocks):
    """The line-faithful C++ restatement of the reference algorithm (oracle, kind='port'), one thread,
    on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    budget_s = args.cpu_seconds
    t0 = time.perf_counter()
    hets = 0
    used = 0
    counters = []
    results = []
    for blk in blocks:
        h1, h2, st, ctr = oracle_ffi.oracle_solve(blk)
        hets += blk.n_variants
        used += 1
        counters.append(ctr)
        results.append((h1, h2, st))
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": hets / dt, "unit": "hets/s", "cores": 1, "kind": "port",
            "sample": f"first {used} block(s) of the same batch ({hets} hets), single thread, {dt:.1f}s"}, counters, results


def usable_cores():
    """Cores this process may actually use: affinity mask, clipped by the cgroup CPU quota, at most 32 threads (each
    thread finishes the block it started, so more threads than cores would overrun the time budget)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:   # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, min(n, 32))


def cpu_baseline_all_cores(args, blocks, skip):
    """The same restatement on every host core: independent blocks on independent threads, as the reference's own
    worker pool does (main.rs:385). ctypes releases the GIL for the duration of each oracle call."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import oracle_ffi
    n_thr = max(1, min(usable_cores(), len(blocks) - skip))
    budget_s = args.cpu_seconds * 0.6
    done = [0] * n_thr
    t0 = time.perf_counter()

    def work(t):
        for blk in blocks[skip + t::n_thr]:
            oracle_ffi.oracle_solve(blk)
            done[t] += blk.n_variants
            if time.perf_counter() - t0 > budget_s:
                break

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "hets/s", "cores": n_thr, "kind": "port",
            "sample": f"{sum(done)} hets of the same batch, one block per thread at a time on {n_thr} threads, {dt:.1f}s"}


def wfa_secondary(device_id):
    """Second line item (not `value`): the graph-WFA allele assignment that builds the matrix rows (SURVEY.md §8d 'WFA
    synthetic': 17-kb reads over 24 het + 8 hom variants, 0.4 % noise), one hp_wfa_assign_batch call of 4096 reads in
    steady state, with a live parity check of a sample against the oracle."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    from hiphase_amd import _ffi
    from hiphase_amd.wfa_graph import PreparedWfaBatch, make_jobs
    from wfa_util import synth_wfa_job
    base = [synth_wfa_job(1000 + s, ref_len=17000, n_vars=24, n_homs=8, noise=0.004)[0] for s in range(32)]
    specs = [base[i % len(base)] for i in range(4096)]
    pb = PreparedWfaBatch(specs)
    pb.run(device_id=device_id)
    res = pb.run(device_id=device_id)
    call_s, kms = pb.last_call_s, _ffi.lib().hp_last_kernel_ms()
    d = oracle_ffi.oracle()
    ok, t0 = True, time.perf_counter()
    for i in range(4):
        jobs, keep = make_jobs([specs[i]])
        o = _ffi.WfaResult()
        al = np.full(max(1, len(specs[i].hets)), 3, np.uint8)
        d.hpo_wfa_assign(C.byref(jobs[0]), 500, 500, C.byref(o), al.ctypes.data)
        ok = ok and (res[i][0], res[i][1], res[i][2]) == (o.status, o.score, o.n_nodes) and np.array_equal(res[i][3], al[:len(specs[i].hets)])
    cpu = (time.perf_counter() - t0) / 4
    return {"reads": len(specs), "read_len": 17000, "kernel_ms": kms, "kernel_reads_per_s": len(specs) / (kms * 1e-3),
            "c_call_ms": call_s * 1e3, "reads_per_s": len(specs) / call_s, "cpu_oracle_reads_per_s": 1.0 / cpu,
            "parity": {"reads_compared": 4, "bit_identical": bool(ok)}}


def so_sha256():
    import hashlib
    from hiphase_amd import _ffi
    return hashlib.sha256(open(_ffi.LIB_PATH, "rb").read()).hexdigest()


def fatbin_sha256(path=None):
    """sha256 of the library's DEVICE code (the ELF section .hip_fatbin): what a kernel's HBM traffic is a property of. A build that
    differs from the profiled one on the host side only (round 5: a mutex around the teardown calls) runs the same kernels."""
    import hashlib
    import struct
    from hiphase_amd import _ffi
    try:
        b = open(path or _ffi.LIB_PATH, "rb").read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return None
        shoff = struct.unpack_from("<Q", b, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)
        stroff = sh(shstrndx)[4]
        for i in range(shnum):
            name_off, _, _, _, off, size = sh(i)[:6]
            if b[stroff + name_off: b.index(b"\0", stroff + name_off)] == b".hip_fatbin":
                return hashlib.sha256(b[off:off + size]).hexdigest()
    except Exception:
        pass
    return None


def measured_traffic(kernels, per_unit_key, units):
    """HBM bytes per launch from the PMC counters: only a measurement taken on THIS build of the library counts
    (profiles/round6/traffic.json records the sha256 of the .so it was measured on, and of its device code - .hip_fatbin -, which is what must match); otherwise null. `kernels`: names to look
    for, the first one the file holds wins (hp_wfa3_kernel, or hp_wfa2_kernel under HP_WFA_GEN=2)."""
    for rnd in ("round6", "round5", "round4", "round3"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic.json")))
        except Exception:
            continue
        for kernel in ([kernels] if isinstance(kernels, str) else kernels):
            e = tj.get(kernel)
            if not e:
                continue
            if e["so_sha256"] != so_sha256() and not (e.get("fatbin_sha256") and e["fatbin_sha256"] == fatbin_sha256()):
                return None, "stale: measured on another build of libhiphase_gpu.so"   # (neither the library nor its device code match)
            return e[per_unit_key] * units, e["source"]
    return None, None


def block_params(cfg=None):
    from hiphase_amd.blocks import _params
    return _params(2, 1000, 3, cfg, True)


def cpu_whole_path(sset, oracle_out, prm, seconds, threads, order):
    """hpo_solve_block (the C++ restatement of the reference's whole path) over blocks of `sset` in `order` on `threads`
    host threads (independent blocks on independent threads, as the reference's own worker pool runs them, main.rs:385;
    ctypes releases the GIL for each call) until `seconds` are spent or the blocks run out. -> (hets, records, blocks done, s)"""
    import ctypes as C
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    d = oracle_ffi.oracle()
    lock = threading.Lock()
    state = {"next": 0, "hets": 0, "records": 0, "done": [], "err": None}
    t0 = time.perf_counter()

    def work():
        while True:
            with lock:
                k = state["next"]
                if k >= len(order) or time.perf_counter() - t0 > seconds or state["err"]:
                    return
                state["next"] = k + 1
            b = order[k]
            rc = d.hpo_solve_block(C.byref(sset.inputs[b]), C.byref(prm), C.byref(oracle_out.arr[b]))
            with lock:
                if rc != 0:
                    state["err"] = (b, rc)
                    return
                state["hets"] += sset.inputs[b].n_hets
                state["records"] += sset.inputs[b].n_records
                state["done"].append(b)

    th = [threading.Thread(target=work) for _ in range(max(1, threads))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if state["err"]:
        raise RuntimeError(f"oracle hpo_solve_block failed on block {state['err'][0]}: {state['err'][1]}")
    return state["hets"], state["records"], state["done"], time.perf_counter() - t0


def drop_in_rates(lib, sets, prm, args):
    """The per-block entries as HiPhase's worker pool would drive them (reference src/main.rs:326-462), same block sets as the
    headline, every block handed over ON ITS OWN: (a) asynchronous - hp_block_submit / hp_block_wait with 40 x 64 = 2 560 blocks in
    flight, the reference's own job slots (main.rs:328); (b) blocking - 64 threads each in hp_solve_blocks(1, ...), the one-call-site
    patch. Behind both the library merges what is in flight into sets for the per-device pipelines. Upload included."""
    import ctypes as C
    import threading
    from hiphase_amd import _ffi
    n_sets = len(sets)
    outs = [s.outputs() for s in sets]
    res = {"threads": 64, "in_flight_async": 2560}

    def blocks_of(passes):
        for k in range(passes):
            s, o = sets[k % n_sets], outs[k % n_sets]
            for b in range(s.n):
                yield s, o, b

    def run_async(passes):
        pending, hets = [], 0
        t0 = time.perf_counter()
        for s, o, b in blocks_of(passes):
            if len(pending) >= 2560:
                _ffi.check(lib.hp_block_wait(pending.pop(0)))
            t = C.c_uint64(0)
            _ffi.check(lib.hp_block_submit(1, C.byref(s.inputs[b]), C.byref(prm), C.byref(o.arr[b]), -1, C.byref(t)))
            pending.append(t.value)
            hets += s.inputs[b].n_hets
        for t in pending:
            _ffi.check(lib.hp_block_wait(t))
        return hets, time.perf_counter() - t0

    def run_blocking(passes):
        work = list(blocks_of(passes))
        lock, state = threading.Lock(), {"next": 0, "err": None}

        def body():
            while True:
                with lock:
                    k = state["next"]
                    if k >= len(work) or state["err"]:
                        return
                    state["next"] = k + 1
                s, o, b = work[k]
                rc = lib.hp_solve_blocks(1, C.byref(s.inputs[b]), C.byref(prm), C.byref(o.arr[b]), -1)
                if rc != 0:
                    state["err"] = rc
                    return

        th = [threading.Thread(target=body) for _ in range(64)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        if state["err"]:
            raise RuntimeError(f"hp_solve_blocks failed: {state['err']}")
        return sum(w[0].inputs[w[2]].n_hets for w in work), time.perf_counter() - t0

    run_async(2)                                  # (starts the dispatcher's pipelines, sizes their buffers)
    h, dt = run_async(max(6, args.steps // 2))
    res["async_hets_per_s"] = h / dt
    run_blocking(1)
    h, dt = run_blocking(3)
    res["blocking_hets_per_s"] = h / dt
    res["note"] = ("every block its own call, merged behind the call into sets for the per-device pipeline; async = hp_block_submit / hp_block_wait "
                   "(the reference's 40 x threads job slots in flight), blocking = 64 threads in hp_solve_blocks(1, ..., -1)")
    return res


def drop_in_rates_cpp(args):
    """The per-block entries measured from C++ threads (tests/cpp/dispatch_test, built by __graft_entry__.build()): 64 std::threads,
    the headline's block-size mix, every block its own call - blocking hp_solve_blocks(1, ..., -1) and hp_block_submit / hp_block_wait
    with 40 tickets per thread - checked against one hp_solve_blocks call over all blocks. A subprocess BEFORE this process touches the GPU: the
    Python loop this replaces submitted 428 ctypes calls per set from ONE interpreter thread and measured the interpreter."""
    import subprocess
    binp = os.path.join(ROOT, "tests", "cpp", "dispatch_test")
    if not os.path.exists(binp):
        return None
    d, failures = None, []
    for attempt in range(2):   # (one more try after a failed start: seen once, right behind a rocprofv3 counter run on the same box - rc and stderr are kept)
        try:
            r = subprocess.run([binp, "64", str(args.total_hets), str(args.max_block_hets), str(max(4, args.steps // 3))], capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            break
        except Exception as e:   # noqa: BLE001
            failures.append({"error": repr(e), "returncode": getattr(locals().get("r"), "returncode", None), "stderr_tail": (getattr(locals().get("r"), "stderr", "") or "")[-400:]})
            time.sleep(2.0)
    if d is None:
        return {"error": failures[-1]["error"], "attempts": failures}
    return {"threads": d["threads"], "in_flight_async": 40 * d["threads"], "async_hets_per_s": d["async_hets_per_s"], "blocking_hets_per_s": d["pool_hets_per_s"],
            "one_call_hets_per_s": d["one_call_hets_per_s"], "blocks": d["blocks"], "passes": d["passes"], "mismatching_blocks": d["mismatching_blocks"], "failed_calls": d["failed_calls"],
            "measured_by": "tests/cpp/dispatch_test (C++ threads, its own process)", **({"failed_attempts": failures} if failures else {}),
            "note": "every block its own call, merged behind the call into sets for the per-device pipeline; async = hp_block_submit / hp_block_wait (the reference's 40 x threads job slots in flight, main.rs:328), blocking = 64 threads in hp_solve_blocks(1, ..., -1); every block compared with one hp_solve_blocks call over all of them"}


def pinned_h2d_gbs():
    """This box's pinned host -> device copy rate, GB/s (scripts/pcie_probe h2d: 1 GiB, best of three), or None"""
    import subprocess
    binp = os.path.join(ROOT, "scripts", "pcie_probe")
    if not os.path.exists(binp):
        return None
    try:
        r = subprocess.run([binp, "h2d"], capture_output=True, text=True, timeout=120)
        vals = [float(l.split(":")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("H2D 1 GiB pinned")]
        return max(vals) if vals else None
    except Exception:   # noqa: BLE001
        return None


def secondary_own_process(args, which):
    """Secondary lines `hifi_mix` / `deep60`: this very bench with `--hifi` (HiFi-shaped errors: per-read rate lognormal around 0.2 %, a
    tail to 1-4 %, half of the errors homopolymer indels) or `--deep60` (BASELINE.json configs[4]'s shape for the whole path: 60x, 15 % of the
    cells carrying the other haplotype's allele so that the A* frontier prunes, every tandem-repeat het multi-allelic, 20 000 hets a set)
    in a process of its own, before this one touches the GPU - as a second stream inside the headline's process a run read 0.9-1.7 M
    hets/s (a new stream's buffers grow beside the old one's 17 GB of pinned sets) where the same run on its own reads 2.4 M. Its own
    parity count: every block of its first timed set against the oracle."""
    import subprocess
    hets = args.total_hets if which == "hifi" else max(2000, args.total_hets // 3)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--" + which, "--no-resident", "--no-drop-in", "--no-hifi", "--no-deep60", "--no-pcie-probe",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-seconds", str(min(args.cpu_seconds, 3.0)), "--total-hets", str(hets),
           "--max-block-hets", str(args.max_block_hets), "--seq-format", args.seq_format, "--depth", str(args.depth), "--host-memory", args.host_memory]
    if which == "deep60":
        cmd += ["--coverage", "60"]
    if args.no_cpu:
        cmd.append("--no-cpu")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)}
    out = {"value": d["value"], "unit": "hets/s", "steps": d["steps"], "ms_per_step": d["ms_per_step"], "period_ms": d.get("period_ms"), "first_completion_ms": d.get("first_completion_ms"),
           "graph_wfa_kernels_ms": d["kernels"][0]["kernel_ms"], "astar_kernel_ms": d["kernels"][1]["kernel_ms"], "wave_updates_per_read": d["kernels"][0]["wave_updates_per_read"],
           "reads_left_compact_path": d["kernels"][0]["reads_left_compact_path"], "records": d["config"]["records"], "hets_per_step": d["config"]["hets_per_step_per_gpu"],
           "parity": d.get("parity"), "cpu_all_cores_hets_per_s": (d.get("cpu_baseline_all_cores") or {}).get("value"), "fallbacks": d.get("fallbacks"),
           "stage_ms": d.get("stage_ms"), "roofline_pcie_frac": (d.get("roofline_pcie") or {}).get("frac"),
           "measured_by": f"python bench.py --{which} (its own process, before the headline's)"}
    if which == "hifi":
        out["workload"] = "the headline's block mix with HiFi-shaped errors: per-read error rate lognormal (median 0.2 %, sigma 0.8, clamp 4 %), half of the errors homopolymer-run indels, no separate noisy class"
    else:
        out["pruned_solutions"] = d.get("pruned_solutions")
        out["astar_cells_per_het"] = d["kernels"][1].get("cells_per_het")
        out["workload"] = ("BASELINE.json configs[4]'s shape for the whole path (hp_synth_reads_deep60): 60x coverage, 15 % of the cells carry the other haplotype's allele (the A* frontier prunes: "
                           "pruned_solutions > 0), every tandem-repeat het multi-allelic (allele0 itself an ALT: 22 % of the hets), 1 % of the reads past max_edit_distance, blocks up to the 4 165-het cap")
    return out


def host_cores():
    """Hardware threads this process may use (affinity mask, clipped by a cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _thread_cpu():
    """tid -> (thread name, CPU seconds so far) of this process's threads (/proc/self/task/*/stat: utime + stime)"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open(f"/proc/self/task/{tid}/stat").read()
            except OSError:
                continue
            name = raw[raw.index("(") + 1:raw.rindex(")")]
            f = raw[raw.rindex(")") + 2:].split()
            if not name.startswith("hp-"):   # not one of the library's: the caller's own thread, or one the HIP runtime started
                name = "caller (main thread)" if tid == str(os.getpid()) else "unnamed (HIP runtime / interpreter threads)"
            out[tid] = (name, (int(f[11]) + int(f[12])) / tick)
    except OSError:
        pass
    return out


def _cgroup_cpu_stat():
    """cpu.stat of the container's cgroup (v2), {} if there is none: nr_throttled / throttled_usec tell whether a CPU quota froze the process"""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            return {k: int(v) for k, v in (l.split() for l in f if len(l.split()) == 2)}
    except OSError:
        return {}


def main_path(args, rank, world, local_rank, dist, backend):
    """Whole-path workload, streamed: every timed step submits a DIFFERENT block set to hp_blockstream_submit - its reads,
    references and variants cross PCIe inside the timed region - and `value` = hets of all steps / wall time from the first
    submit to the last result. The sets are generated (C generator, C layout) before the clock starts."""
    import ctypes as C
    import numpy as np
    from hiphase_amd import _ffi
    from hiphase_amd.synth_sets import SynthSet, default_spec
    lib = _ffi.lib()
    fmt = _ffi.SEQ_BAM4 if args.seq_format == "bam4" else _ffi.SEQ_ASCII
    # Two measurements by other PROCESSES, taken before this one creates its HIP context and queues (a second process on a GPU whose
    # hardware queues this one already holds is time-sliced against them: the same dispatch_test read 17 k instead of 700 k hets/s
    # when it ran after the timed region) - outside the timed region either way: the box's pinned host-to-device rate for
    # `roofline_pcie`, and the per-block entries driven by 64 C++ threads for `drop_in`.
    pre_h2d = pinned_h2d_gbs() if (rank == 0 and world == 1 and not args.no_pcie_probe) else None
    pre_drop_in = drop_in_rates_cpp(args) if (rank == 0 and world == 1 and not args.no_drop_in) else None
    secondary_ok = rank == 0 and world == 1 and not args.hifi and not args.deep60 and not args.replay and not args.inproc
    pre_hifi = secondary_own_process(args, "hifi") if (secondary_ok and not args.no_hifi) else None
    pre_deep60 = secondary_own_process(args, "deep60") if (secondary_ok and not args.no_deep60) else None
    capture = None
    if args.replay:   # a .hpbr capture of real blocks: the same blocks every step, still crossing PCIe every step
        from hiphase_amd.synth_sets import Capture
        capture = Capture(args.replay)
        if world > 1:
            raise SystemExit("--replay of a .hpbr capture runs on one GPU (the multi-GPU replay is hp_solve_blocks(device_id=-1))")
    # (N ranks on one host: 1.3 GB of host memory per generated set and rank - fewer distinct sets each, never fewer than the stream
    # holds in flight plus one; a set that comes round again still crosses PCIe)
    n_local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    per_rank_sets = args.distinct_sets if n_local == 1 else max(args.depth + 1, min(args.distinct_sets, 48 // n_local))
    n_sets = 1 if capture else max(1, min(args.steps + max(args.warmup, args.depth + 1), per_rank_sets))
    gen_threads = max(2, min(32, host_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    t_gen = time.perf_counter()
    sets = [capture] if capture else []
    for k in range(0 if capture else n_sets):   # rank r, set k: seed + 1000 r + k (disjoint over ranks and steps)
        over = {}
        for kv in args.spec:   # e.g. --spec edit_noise=0.003 --spec frac_sv=0
            key, val = kv.split("=", 1)
            over[key] = float(val)
        sets.append(SynthSet(default_spec(lib, hifi=args.hifi, deep60=args.deep60, **dict(dict(seed=args.seed + 1000 * rank + k, total_hets=args.total_hets, max_block_hets=args.max_block_hets,
                                                           coverage=float(args.coverage), seq_format=fmt, threads=gen_threads), **over))))
    n_pinned = 0
    if args.host_memory == "pinned" and not capture:   # the records' bases gathered in hp_host_alloc memory: read in place, nothing staged
        for s_ in sets:
            try:
                s_.relocate_pinned()
                n_pinned += 1
            except _ffi.HpError:   # (the box would not pin another gigabyte: that set stays where it is and is staged)
                break
    outs = [s.outputs() for s in sets]
    t_gen = time.perf_counter() - t_gen
    prm = block_params()
    if capture and capture.n:
        prm = capture.params[0]   # (the parameters the blocks were captured with)
    st = C.c_int(0)
    # --inproc: ONE process, one pipeline per visible device behind one stream (device_id = -1) - the form a single HiPhase process
    # on a multi-GPU node uses; HP_STREAM_DEVICES=n stands n pipelines on a box with fewer GPUs
    stream = stream0 = lib.hp_blockstream_create(C.byref(prm), -1 if args.inproc else local_rank, args.depth, C.byref(st))
    if not stream:
        raise SystemExit(f"hp_blockstream_create failed: {st.value} {lib.hp_last_error().decode()}")
    n_pipes = lib.hp_blockstream_devices(stream)
    in_flight_cap = args.depth * n_pipes

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def run(first, count, stream=None, sets=sets, outs=outs, done_at=None):
        """submits sets first .. first + count - 1 (cycling over the generated ones), at most `depth` in flight; -> per-set stage_ms, work.
        done_at: list that receives the time each set's wait returned (the stream's period = the median interval between them)"""
        stream = stream or stream0
        pending, stages, works = [], [], []
        ms, work = (C.c_double * 16)(), (C.c_uint64 * 8)()

        def wait_oldest():
            _ffi.check(lib.hp_blockstream_wait(stream, pending.pop(0), ms, work))
            if done_at is not None:
                done_at.append(time.perf_counter())
            stages.append(list(ms))
            works.append(list(work))

        for k in range(first, first + count):
            if len(pending) >= in_flight_cap:
                wait_oldest()
            i = k % len(sets)
            t = C.c_uint64(0)
            _ffi.check(lib.hp_blockstream_submit(stream, sets[i].n, sets[i].inputs, outs[i].arr, C.byref(t)))
            pending.append(t.value)
        while pending:
            wait_oldest()
        return stages, works

    def period_of(done_at):
        """MEAN interval between consecutive completions, ms = (last completion - first completion) / (sets - 1): the stream's steady-state
        period (ms_per_step = wall / steps also holds the first set's way through an empty stream, a fifth of a 20-step run). The mean, not
        the median: with two threads in the last stage the completions come in bursts (a 4 ms interval, then a 36 ms one), and the median
        of 19 such intervals read 17.6 ms - below what the set's bytes need on the PCIe link (round 5: roofline_pcie.frac 1.06)."""
        return 1e3 * (done_at[-1] - done_at[0]) / (len(done_at) - 1) if len(done_at) > 1 else None

    def period_median_of(done_at):
        gaps = sorted(b - a for a, b in zip(done_at, done_at[1:]))
        return 1e3 * gaps[len(gaps) // 2] if gaps else None

    # (untimed warm-up: at least depth + 1 sets whatever --warmup says - every slot of the stream must have sized its device and
    # pinned buffers once, hipMalloc / hipHostMalloc wait for the whole device - reported as warmup_run)
    warm = max(args.warmup, in_flight_cap + 1)
    run(0, warm)
    sync_all()
    cg0, cpu0, th0 = _cgroup_cpu_stat(), time.process_time(), _thread_cpu()
    t0 = time.perf_counter()
    done_at = []
    stages, works = run(warm, args.steps, done_at=done_at)      # every wait returns with that set's results in the caller's buffers
    sync_all()
    elapsed = time.perf_counter() - t0
    cg1, cpu1, th1 = _cgroup_cpu_stat(), time.process_time(), _thread_cpu()
    by_thread = {}
    for tid, (name, sec) in th1.items():   # the library names its threads: hp-s<stage>, their pool workers ...w, helpers ...h
        d = sec - th0.get(tid, (name, 0.0))[1]
        if d > 0:
            by_thread[name] = by_thread.get(name, 0.0) + d
    if os.environ.get("HP_BENCH_THREAD_DUMP"):   # which threads are the unnamed ones? (tid, CPU share, what the kernel says they wait in)
        rows = []
        for tid, (name, sec) in th1.items():
            d = sec - th0.get(tid, (name, 0.0))[1]
            if d / max(elapsed, 1e-9) >= 0.01:
                def rd(f):
                    try:
                        return open(f"/proc/self/task/{tid}/{f}").read().strip().replace("\n", " | ")[:400]
                    except OSError as e:
                        return f"({e.strerror})"
                rows.append((d / elapsed, tid, name, rd("comm"), rd("wchan"), rd("stack")))
        for r in sorted(rows, reverse=True):
            print("[bench] thread %s %-44s comm %-16s cpu %.3f wchan %s stack %s" % (r[1], r[2], r[3], r[0], r[4], r[5]), file=sys.stderr)
    host_cpu = {"process_cpu_s_per_wall_s": (cpu1 - cpu0) / elapsed if elapsed > 0 else None,
                "by_thread_name_cpu_s_per_wall_s": {k: round(v / elapsed, 3) for k, v in sorted(by_thread.items(), key=lambda kv: -kv[1]) if v / elapsed >= 0.005},
                "cgroup_throttled_periods": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0) if cg0 and cg1 else None,
                "cgroup_throttled_ms": (cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3 if cg0 and cg1 else None,
                "note": "host side of the timed region: CPU seconds the process used per second of wall time, and how often the container's CPU quota froze its threads (cpu.stat)"}
    hets_timed = sum(sets[k % n_sets].info["hets"] for k in range(warm, warm + args.steps))
    per_rank = None
    if dist is not None:
        from hiphase_amd.shard import gather_per_rank, max_over_ranks
        dev_ = "cuda" if backend == "nccl" else "cpu"
        # every rank's own clock, hets and host share, so that a scaling run explains itself: a rank that ran slower than the others
        # (host contention: N generators and N x 3.4 CPU-seconds per second on one host) shows here, not only in the maximum
        rows_ = gather_per_rank(dist, [elapsed, hets_timed, host_cpu["process_cpu_s_per_wall_s"] or 0.0, float(lib.hp_runtime_wait_mode())], device=dev_)
        per_rank = [{"rank": r_, "elapsed_s": e_, "hets_per_s": h_ / e_ if e_ > 0 else None, "host_cpu_s_per_wall_s": c_, "wait_mode": int(w_)} for r_, (e_, h_, c_, w_) in enumerate(rows_)]
        elapsed = max_over_ranks(dist, elapsed, device=dev_)   # timing only; no block data crosses ranks
    if rank == 0:
        st_mean = np.mean(np.asarray(stages), axis=0)
        work = dict(zip(("wfa_reads", "wfa_read_bytes", "wfa_node_bytes", "wfa_updates", "astar_cells", "astar_evals", "hets", "rows"),
                        np.mean(np.asarray(works, dtype=np.float64), axis=0)))
        info = sets[warm % n_sets].info
        k_wfa_ms, k_astar_ms = st_mean[8], st_mean[9]
        # graph-WFA kernels: algorithmic bytes = read bases + bytes of the traversed graph nodes + 8 B per (node, diagonal) wave update
        # (SURVEY.md 8d), counted on the device for the reads the compact kernels aligned. kernel_ms: the three graph-size
        # instantiations run concurrently on three streams; HIP events around the launch set give their span per set
        b_wfa = work["wfa_read_bytes"] + work["wfa_node_bytes"] + 8 * work["wfa_updates"]
        b_astar = BYTES_PER_CELL * work["astar_cells"]
        gen2 = os.environ.get("HP_WFA_GEN", "3").startswith("2")
        kname = "hp_wfa2_kernel" if gen2 else "hp_wfa3_kernel"
        k_wfa = {"kernel": f"hp::{kname}<8,2> + <8,4> + <16,8> (concurrent; span of the launch set - the largest class's tail included -, mean over the timed sets)", "bound": "hbm",
                 "kernel_ms": k_wfa_ms, "algorithmic_bytes_per_launch": b_wfa, "achieved": b_wfa / (k_wfa_ms * 1e-3) / 1e9 if k_wfa_ms > 0 else 0.0,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "reads": work["wfa_reads"], "reads_per_s": work["wfa_reads"] / (k_wfa_ms * 1e-3) if k_wfa_ms > 0 else 0.0,
                 "bytes_per_read": b_wfa / max(1.0, work["wfa_reads"]), "wave_updates_per_read": work["wfa_updates"] / max(1.0, work["wfa_reads"]),
                 "reads_left_compact_path": float(np.mean([sets[k % n_sets].info["records"] for k in range(warm, warm + args.steps)])) - work["wfa_reads"]}
        k_wfa["frac"] = k_wfa["achieved"] / HBM_PEAK_GBS
        k_wfa["traffic"], k_wfa["traffic_source"] = measured_traffic(["hp_wfa2_kernel"] if gen2 else ["hp_wfa3_kernel"], "bytes_per_read", work["wfa_reads"])
        k_astar = {"kernel": "hp::hp_astar_kernel", "bound": "hbm", "kernel_ms": k_astar_ms, "algorithmic_bytes_per_launch": b_astar,
                   "achieved": b_astar / (k_astar_ms * 1e-3) / 1e9 if k_astar_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "cells_per_het": work["astar_cells"] / max(1, info["hets"])}
        k_astar["frac"] = k_astar["achieved"] / HBM_PEAK_GBS
        # (kernel_ms above covers hp_astar_kernel AND the segment-parallel heuristic hp_heur_seg_kernel: so does the traffic - 0.5 + 13.6 KB per het in round 4)
        k_astar["traffic"], k_astar["traffic_source"] = measured_traffic("hp_astar_kernel", "bytes_per_het", info["hets"])
        t_seg, _src = measured_traffic("hp_heur_seg_kernel", "bytes_per_het", info["hets"])
        if k_astar["traffic"] is not None and t_seg is not None:
            k_astar["traffic"] += t_seg
            k_astar["kernel"] = "hp::hp_astar_kernel + hp::hp_heur_seg_kernel (kernel_ms and traffic: both)"
        for k in (k_wfa, k_astar):   # what actually crossed the HBM interface, next to the algorithmic figure
            k["traffic_gbs"] = k["traffic"] / (k["kernel_ms"] * 1e-3) / 1e9 if k["traffic"] and k["kernel_ms"] > 0 else None
            k["traffic_frac"] = k["traffic_gbs"] / HBM_PEAK_GBS if k["traffic_gbs"] else None
        # the dominant kernel = the one that holds the device: the graph-WFA launch set (53.6 % of all kernel time in
        # profiles/round5/path_kernel_stats.csv, 2.0e9 busy cycles a set, 92 % of the wavefront slots while it runs) - not the A* figure
        # beside it, which is the LENGTH of a latency chain of a few hundred wavefronts on two streams (18.7 % of the kernel time, 5.5e8
        # busy cycles) and can read longer than the launch set's span (round 5: 19.7 against 18.4 ms). Both are in `kernels`.
        dom = k_wfa
        ms_step = elapsed / args.steps * 1e3
        period_ms = period_of(done_at)
        h2d = pre_h2d   # (measured by a subprocess before this process touched the GPU)
        pcie = None
        if h2d and period_ms:
            need_ms = st_mean[10] / (h2d * 1e9) * 1e3
            pcie = {"bound": "pcie", "bytes_per_step": st_mean[10], "peak": h2d, "unit": "GB/s", "achieved": st_mean[10] / (period_ms * 1e-3) / 1e9,
                    "frac": need_ms / period_ms, "floor_ms_per_step": need_ms, "hets_per_s_at_the_floor": info["hets"] / (need_ms * 1e-3),
                    "note": "what bounds the PATH: a set's bytes over the box's own pinned host-to-device rate (scripts/pcie_probe, its own process, before the timed region) against the stream's period; `roofline` is the dominant kernel against HBM"}
        out = {
            "metric": "het variants phased/sec, whole path, streamed (every step a new block set: records over PCIe -> graph-WFA -> rows -> A* -> span counts / haplotags)",
            "value": hets_timed * world / elapsed,
            "unit": "hets/s", "n_gpus": world if not args.inproc else n_pipes, "pipelines": n_pipes, "inproc": bool(args.inproc), "steps": args.steps, "warmup": args.warmup, "warmup_run": warm,
            "ms_per_step": ms_step, "period_ms": period_ms, "period_median_ms": period_median_of(done_at), "first_completion_ms": (done_at[0] - t0) * 1e3 if done_at else None,
            "completion_intervals_ms": [round((b - a) * 1e3, 1) for a, b in zip(done_at, done_at[1:])],
            "first_sets_stage_ms": [[round(x, 1) for x in st_] for st_ in stages[:3]], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u64", "data": "synthetic", "host_cpu": host_cpu, "wait_mode": int(lib.hp_runtime_wait_mode()), "per_rank": per_rank, "roofline_pcie": pcie,
            "config": {"workload": (f"synthetic read-bearing WGS-like block sets, one NEW set per step and GPU through hp_blockstream_* ({args.depth} sets in flight): "
                                    f"{info['blocks']} blocks, {info['hets']} hets (lognormal block sizes, median 15, max {info['max_block_hets']}), "
                                    f"{info['records']} records of {info['read_bases'] / max(1, info['records']):.0f} b mean at {args.coverage}x "
                                    f"(0.5% edit noise: sub / ins / del; 0.3% of the reads at 5%: they exceed max_edit_distance; 2% supplementary), "
                                    f"het + hom calls SNV .85 / indel .12 / SV .01 / tandem repeat .02, reads handed over as {args.seq_format}" + (", gathered in hp_host_alloc memory" if n_pinned else "")),
                       "blocks": info["blocks"], "hets_per_step_per_gpu": info["hets"], "records": info["records"], "read_bases": info["read_bases"],
                       "distinct_sets": n_sets, "depth": args.depth, "seq_format": args.seq_format, "host_memory": (f"hp_host_alloc arena ({n_pinned} of {len(sets)} sets)" if n_pinned else "pageable"),
                       "host_to_device_bytes_per_step": st_mean[10], "host_to_device_gbs": st_mean[10] / (ms_step * 1e-3) / 1e9,
                       "min_queue_size": 1000, "queue_increment": 3, "max_edit_distance": 500, "wfa_prune_distance": 500,
                       "generate_s": round(t_gen, 2), "spec_overrides": args.spec},
            "stage_ms": {"overlaps_layout_host": st_mean[0], "staging_pcie_expand": st_mean[1], "graph_wfa": st_mean[2], "fallback_rows_collapse_host": st_mean[3],
                         "astar_pack_upload": st_mean[4], "astar_solve": st_mean[5], "postprocess_outputs": st_mean[6], "latency_submit_to_done": st_mean[7],
                         "graph_wfa_kernels": st_mean[8], "astar_kernel": st_mean[9], "waiting_between_stages": st_mean[11],
                         "stage1_wall": st_mean[12], "stage2_wall": st_mean[13], "stage3_wall": st_mean[14], "stage4_wall": st_mean[15],
                         "note": "means over the timed sets, ms. graph_wfa / stage2_wall = the alignment stage's THREAD time: it queues a set's kernels and moves on (what it waits for is the previous set's class kernels); graph_wfa_kernels = the launch set's span on the device (HIP events). fallback_rows_collapse_host / stage3_wall include the wait for the set's first collection and for its late results. overlaps_layout_host = the layout stage incl. the host-only half of the sequence layout; staging_pcie_expand / stage1_wall = tables + DMA"},
            "roofline": {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_gbs", "traffic_frac", "kernel", "kernel_ms", "algorithmic_bytes_per_launch")},
            "kernels": [k_wfa, k_astar],
        }
        lib.hp_blockstream_destroy(stream)
        stream = None
        if world == 1 and not args.no_resident:
            # secondary: the same path over ONE set whose inputs are already in HBM (hp_blockset_solve again and again): what the
            # pipeline would do if PCIe and the host stages were free and nothing overlapped
            s0 = sets[warm % n_sets]
            stc = C.c_int(0)
            t_up = time.perf_counter()
            bs = lib.hp_blockset_create(s0.n, s0.inputs, C.byref(prm), local_rank, C.byref(stc))
            t_up = time.perf_counter() - t_up
            if bs:
                ms8 = (C.c_double * 8)()
                ro = s0.outputs()
                for _ in range(2):
                    _ffi.check(lib.hp_blockset_solve(bs, ro.arr, ms8))
                t1 = time.perf_counter()
                reps = max(3, args.steps // 2)
                for _ in range(reps):
                    _ffi.check(lib.hp_blockset_solve(bs, ro.arr, ms8))
                dt = (time.perf_counter() - t1) / reps
                lib.hp_blockset_destroy(bs)
                out["resident"] = {"hets_per_s": s0.info["hets"] / dt, "ms_per_step": dt * 1e3, "layout_upload_ms": t_up * 1e3,
                                   "note": "hp_blockset_solve over one resident set (inputs in HBM, no overlap between stages)"}
                out["streamed_over_resident"] = out["value"] / out["resident"]["hets_per_s"]
        if world == 1 and not args.no_drop_in:
            out["drop_in"] = pre_drop_in or drop_in_rates(lib, sets, prm, args)
        if capture:
            out["data"] = "replay of " + os.path.basename(args.replay)
            out["config"]["workload"] = f"replay of the read-bearing capture {os.path.basename(args.replay)}: {info['blocks']} blocks, {info['hets']} hets, {info['records']} records, streamed again every step"
            have = [b for b in range(capture.n) if capture.expected[b].status != -2 ** 31]
            ok = all(bool(lib.hp_block_output_equal(C.byref(capture.inputs[b]), C.byref(outs[0].arr[b]), C.byref(capture.expected[b]))) for b in have)
            out["parity_vs_capture"] = {"blocks_compared": len(have), "of": capture.n, "bit_identical": bool(ok)}
        if not args.no_cpu and world == 1:
            # CPU leg + parity, on the first timed set: the oracle's whole path (hpo_solve_block) on every host core over ALL its
            # blocks - that is also the parity check of every block - and on one thread over a random sample of them
            i0 = warm % n_sets
            s0, gpu_out = sets[i0], outs[i0]
            cores = host_cores()
            rng = np.random.default_rng(12345)
            order = sorted(range(s0.n), key=lambda b: -s0.inputs[b].n_records)   # largest first: the tail of the all-cores run is one block
            oo = s0.outputs()
            h, r, done, dt = cpu_whole_path(s0, oo, prm, 10.0 * args.cpu_seconds, cores, order)
            out["cpu_baseline_all_cores"] = {"value": h / dt, "unit": "hets/s", "cores": cores, "kind": "port",
                                             "sample": f"{len(done)} of {s0.n} blocks of the first timed set ({h} hets, {r} records) through the whole path on the C++ restatement, "
                                                       f"one block per thread at a time on {cores} threads, {dt:.1f}s"}
            ok = all(gpu_out.equal(oo, b) for b in done)
            out["parity"] = {"blocks_compared": len(done), "of": s0.n, "hets_compared": h, "bit_identical": bool(ok),
                             "what": "every field hp_solve_blocks fills: segments (alleles, quals, regions), haplotypes, PhaseStats, span counts, haplotags, read statistics, edit distances"}
            o1 = s0.outputs()
            h1, r1, done1, dt1 = cpu_whole_path(s0, o1, prm, args.cpu_seconds, 1, [int(b) for b in rng.permutation(s0.n)])
            out["cpu_baseline"] = {"value": h1 / dt1, "unit": "hets/s", "cores": 1, "kind": "port",
                                   "sample": f"{len(done1)} blocks drawn at random from the first timed set ({h1} hets, {r1} records) through the whole path on the C++ restatement, single thread, {dt1:.1f}s"}
            out["fallbacks"] = {"local_aligned": int(sum(gpu_out.arr[b].local_aligned for b in range(s0.n))), "global_aligned": int(sum(gpu_out.arr[b].global_aligned for b in range(s0.n)))}
            out["pruned_solutions"] = int(sum(gpu_out.arr[b].stats.pruned_solutions for b in range(s0.n) if gpu_out.arr[b].status == 0))   # (PhaseStats of the first timed set: > 0 = the A* frontier pruned, astar_phaser.rs:564-585)
        if pre_hifi is not None:
            out["hifi_mix"] = pre_hifi
        if pre_deep60 is not None:
            out["deep60"] = pre_deep60
        print(json.dumps(out), flush=True)
    if stream:
        lib.hp_blockstream_destroy(stream)
    if dist is not None:
        dist.destroy_process_group()


def _blocking_sync_through_torchs_runtime(device):
    """hipSetDeviceFlags(hipDeviceScheduleBlockingSync) on the rank's device (None: every device) through the HIP runtime torch
    loaded, before anything in the process has initialised a device. Best effort: a refusal leaves the runtime's default (spinning)."""
    import ctypes as _C
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    try:
        hip = _C.CDLL(path if os.path.exists(path) else "libamdhip64.so")
        n = _C.c_int(0)
        if hip.hipGetDeviceCount(_C.byref(n)) != 0:
            return
        for d in ([device] if device is not None else range(n.value)):
            if 0 <= d < n.value and hip.hipSetDevice(d) == 0:
                hip.hipSetDeviceFlags(4)   # hipDeviceScheduleBlockingSync
        hip.hipSetDevice(device if device is not None and device < n.value else 0)
    except OSError:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6, help="untimed steps first (path workload: at least depth + 1, so that every slot of the stream has sized its buffers)")
    ap.add_argument("--workload", choices=["path", "c2", "wgs"], default="path")
    ap.add_argument("--total-hets", type=int, default=60000, help="path workload: hets per GPU and step")
    ap.add_argument("--max-block-hets", type=int, default=4165, help="largest block HiPhase reports on HG002 (docs/user_guide.md:258)")
    ap.add_argument("--seq-format", choices=["bam4", "ascii"], default="bam4", help="path workload: how the reads are handed over")
    ap.add_argument("--depth", type=int, default=7, help="path workload: block sets in flight in the stream (measured: 5 -> 31, 6 -> 28.3-29.4, 7 -> 29.2-30.2, 8 -> 29.3-29.6 ms per step; 160 / 190 / 215 ms from submit to done at 6 / 7 / 8)")
    ap.add_argument("--distinct-sets", type=int, default=16, help="path workload: generated sets (steps + warm-up if fewer; cycled if more are needed)")
    ap.add_argument("--host-memory", choices=["pageable", "pinned"], default="pinned",
                    help="path workload: where the records' bases lie on the host - ordinary memory (staged by the library's host threads) or hp_host_alloc memory (read in place by the device)")
    ap.add_argument("--spec", action="append", default=[], help="path workload: override a field of hp_synth_reads_spec, key=value (repeatable)")
    ap.add_argument("--no-resident", action="store_true", help="path workload: skip the secondary resident (inputs-in-HBM) figure")
    ap.add_argument("--no-drop-in", action="store_true", help="path workload: skip the secondary per-block (drop-in) rates")
    ap.add_argument("--inproc", action="store_true", help="path workload: one process drives every visible device through ONE stream (hp_blockstream_create(device_id = -1)); steps = sets over all devices")
    ap.add_argument("--no-hifi", action="store_true", help="path workload: skip the secondary line on HiFi-shaped errors")
    ap.add_argument("--no-pcie-probe", action="store_true", help="path workload: skip the pinned host-to-device rate probe (roofline_pcie)")
    ap.add_argument("--no-deep60", action="store_true", help="path workload: skip the secondary line on the 60x / conflicting-rows / multi-allelic set (BASELINE.json configs[4]'s shape)")
    ap.add_argument("--deep60", action="store_true", help="path workload: the HEADLINE run on hp_synth_reads_deep60 (use with --coverage 60 --total-hets 20000)")
    ap.add_argument("--hifi", action="store_true", help="path workload: the HEADLINE run on the HiFi-shaped error model instead of uniform 0.5 %% (for profiling it)")
    ap.add_argument("--seed", type=int, default=20250928, help="path workload: seed of the synthetic block mix (rank r uses seed + r)")
    ap.add_argument("--blocks", type=int, default=6144, help="blocks per GPU (24 resident single-wave workgroups per CU x 256 CUs)")
    ap.add_argument("--hets", type=int, default=5000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--span", type=int, default=20)
    ap.add_argument("--error", type=float, default=0.01)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the WGS-like secondary measurement")
    ap.add_argument("--replay", default=None, help=".hpbr capture of real phase blocks (read-bearing: the whole path) or .hpbk (solver matrices: solver stage, strong scaling over ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("HP_BENCH_BACKEND", "nccl")   # "gloo": control-flow test of the N>1 path on a box with fewer GPUs
    # N = 1: the library is the first (and only) user of the HIP runtime in this process and decides how host threads wait for the device
    # (hipDeviceScheduleBlockingSync, hp_runtime_wait_mode). N > 1: torch comes first - its wheel bundles its own libamdhip64 under the
    # same soname, and the library must bind to THAT copy (two HIP runtimes in one process: the second one finds no GPU) - so the
    # wait mode is set through torch's runtime before torch initialises the device; the library then finds it set.
    if world > 1:
        import torch
        import torch.distributed as dist
        _blocking_sync_through_torchs_runtime(local_rank if backend == "nccl" else None)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)

    from hiphase_amd import ResidentBatch, _ffi
    lib = _ffi.lib()
    if lib.hp_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libhiphase_gpu.so has no CPU fallback")

    if args.replay and args.workload == "path" and not args.replay.endswith(".hpbr"):
        args.workload = "c2"   # a .hpbk capture holds solver matrices: replay is a solver-stage run (.hpbr: read-bearing, the whole path)
    if args.workload == "path":
        return main_path(args, rank, world, local_rank, dist, backend)

    blocks = make_blocks(args, rank)
    hets_per_step = sum(b.n_variants for b in blocks)
    t_pack = time.perf_counter()
    rb = ResidentBatch(blocks, device_id=local_rank)   # pack + upload: inputs resident in HBM from here on
    t_pack = time.perf_counter() - t_pack

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        rb.solve()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(rb.solve())       # launches on the batch stream and waits for it (HIP events inside)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        from hiphase_amd.shard import max_over_ranks
        elapsed = max_over_ranks(dist, elapsed, device="cuda" if backend == "nccl" else "cpu")   # timing only; no block data crosses ranks

    res, ctrs, _ = rb.results()
    cells_per_step = sum(c.cells for c in ctrs)
    evals_per_step = sum(c.evals for c in ctrs)
    cyc_heur = sum(c.reserved[0] for c in ctrs) / max(1, len(ctrs))
    cyc_main = sum(c.reserved[1] for c in ctrs) / max(1, len(ctrs))
    # HBM traffic (PMC) cannot be collected from inside this process; the value measured with
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on the same workload is kept under profiles/ and scaled
    # to this launch's hets (see profiles/round1/README.md). null when no measurement matches the workload.
    # (profiles/round2/traffic.json was measured on the whole-path workload, where the blocks' matrices stay in L2: it says
    # nothing about this one - 14.7 KB per het in round 1, profiles/round1/traffic.json, on another build)
    traffic, traffic_src = None, "not measured for this workload on this build"
    out = None
    if rank == 0:
        kavg_ms = sum(kernel_ms) / len(kernel_ms)
        b_alg = BYTES_PER_CELL * cells_per_step
        achieved = b_alg / (kavg_ms * 1e-3) / 1e9
        out = {
            "metric": "het variants phased/sec, A* MEC solver stage only (resident synthetic read-allele matrices)",
            "value": hets_per_step * world * args.steps / elapsed,
            "unit": "hets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": (f"C2 x {args.blocks} blocks/GPU: N={args.hets} hets x R={blocks[0].n_reads} reads "
                                    f"(C={args.coverage}, S={args.span}, e={args.error}, a=0.02), seeds 20250509+i"
                                    if args.workload == "c2" else
                                    f"WGS-like: {args.blocks} blocks/GPU, lognormal sizes (median 15, max 4000), C={args.coverage}"),
                       "min_queue_size": 1000, "queue_increment": 3, "hets_per_step_per_gpu": hets_per_step,
                       "pack_upload_s": round(t_pack, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "hp::hp_astar_kernel", "kernel_ms": kavg_ms,
                         "algorithmic_bytes_per_launch": b_alg, "cells_per_het": cells_per_step / hets_per_step,
                         "evals_per_het": evals_per_step / hets_per_step,
                         "mean_wave_cycles_heuristic": cyc_heur, "mean_wave_cycles_main": cyc_main},
        }
        if args.replay:
            out["scaling"] = "strong"
            out["config"]["workload"] = f"replay of {os.path.basename(args.replay)}: {len(blocks)} captured blocks on rank 0"
            exp = [e for e in getattr(args, "replay_expected", []) if e is not None]
            if exp:
                ok = all((r.haplotype_1 == e[0]).all() and (r.haplotype_2 == e[1]).all() and r.statistics.as_tuple() == tuple(e[2])
                         for r, e in zip(res, getattr(args, "replay_expected")) if e is not None)
                out["parity_vs_capture"] = {"blocks_compared": len(exp), "bit_identical": bool(ok)}
        if not args.no_cpu and world == 1:   # the CPU leg is timed at N=1 only
            cb, octr, ores = cpu_baseline(args, blocks)
            out["cpu_baseline"] = cb
            if usable_cores() > 1 and len(blocks) > len(ores) + 1:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, blocks, len(ores))
            # the oracle doubles as a live parity check on the sampled blocks
            ok = all((r.haplotype_1 == o[0]).all() and (r.haplotype_2 == o[1]).all() and r.statistics.as_tuple() == o[2]
                     for r, o in zip(res, ores))
            ok = ok and all(c.as_tuple() == o for c, o in zip(ctrs, octr))
            out["parity"] = {"blocks_compared": len(ores), "bit_identical": bool(ok)}
        if world == 1 and args.workload == "c2" and not args.replay and not args.no_secondary:
            # secondary line item (not `value`): a heavy-tailed WGS-like block-size mix (median 15, max 4000 hets,
            # docs/user_guide.md:257) where the critical path, not throughput, is what counts
            import copy
            a2 = copy.copy(args)
            a2.workload, a2.blocks = "wgs", 12000
            b2 = make_blocks(a2, 0)
            rb2 = ResidentBatch(b2, device_id=local_rank)
            rb2.solve()
            ms2 = min(rb2.solve() for _ in range(3))
            rb2.close()
            h2 = sum(b.n_variants for b in b2)
            out["secondary_wgs_like"] = {"blocks": len(b2), "hets": h2, "kernel_ms": ms2, "hets_per_s": h2 / (ms2 * 1e-3),
                                         "note": "lognormal block sizes (median 15, max 4000); segment-parallel heuristic on"}
            out["secondary_graph_wfa"] = wfa_secondary(local_rank)
        print(json.dumps(out), flush=True)
    rb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
