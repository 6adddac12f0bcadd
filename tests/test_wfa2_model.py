"""CPU pin of the round-2 graph-WFA design (tests/cpp/wfa2_model.cpp): the device graph builder compiled for the
host and the compact wavefront formulation (per-round hull arenas, injection list, CAPPED-diagonal set instead of the
max_wavefronts map) must reproduce the oracle's (status, score, node count, alleles) on the reference's golden graphs
and on randomised jobs, whenever they stay inside the kernel's capacity limits."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden
from hiphase_amd import _ffi
from hiphase_amd.wfa_graph import make_jobs
from oracle_ffi import oracle
from wfa_util import spec_from_golden, synth_wfa_job, _Rng

HERE = os.path.dirname(os.path.abspath(__file__))
G = load_golden("wfa_graph.json")


def model():
    so = os.path.join(HERE, "cpp", "libwfa2_model.so")
    src = os.path.join(HERE, "cpp", "wfa2_model.cpp")
    hdr = os.path.join(HERE, "..", "hiphase_amd", "csrc", "hp_wfa2_dev.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
    d = C.CDLL(so)
    d.w2m_wfa_assign.restype = C.c_int
    d.w3m_wfa_assign.restype = C.c_int
    return d


GEN = 2   # which formulation compare() holds to the oracle: 2 = per-node hull arenas (hp_wfa2_kernel), 3 = flat sorted slot lists (hp_wfa3_kernel)


def compare(specs, prune, max_ed, m=None, d=None, gen=None):
    m = m or model()
    d = d or oracle()
    assign = m.w3m_wfa_assign if (gen or GEN) == 3 else m.w2m_wfa_assign
    paths = [0, 0, 0]
    for i, spec in enumerate(specs):
        jobs, keep = make_jobs([spec])
        pr = (2 ** 64 - 1) if prune in (0, None) else prune
        o1, o2 = _ffi.WfaResult(), _ffi.WfaResult()
        a1 = np.full(max(1, len(spec.hets)), 3, np.uint8)
        a2 = np.full(max(1, len(spec.hets)), 3, np.uint8)
        assert d.hpo_wfa_assign(C.byref(jobs[0]), C.c_uint64(pr), C.c_uint64(max_ed), C.byref(o1), C.c_void_p(a1.ctypes.data)) == 0
        path = C.c_int(0)
        rc = assign(C.byref(jobs[0]), C.c_uint64(pr), C.c_uint64(max_ed), C.byref(o2), C.c_void_p(a2.ctypes.data), C.byref(path))
        assert rc == 0, (i, rc)
        paths[path.value] += 1
        if path.value != 0:
            continue
        assert (o1.status, o1.score, o1.n_nodes) == (o2.status, o2.score, o2.n_nodes), (i, (o1.status, o1.score, o1.n_nodes), (o2.status, o2.score, o2.n_nodes))
        assert np.array_equal(a1, a2), (i, a1.tolist(), a2.tolist())
    return paths


@pytest.mark.parametrize("gen", [2, 3])
@pytest.mark.parametrize("case", [c for c in G["variant_built"] if c["queries"]], ids=lambda c: c["name"])
def test_golden_variant_graphs(case, gen):
    specs = [spec_from_golden(case, read=bytes(q["seq"])) for q in case["queries"]]
    paths = compare(specs, 0, 1000, gen=gen)
    assert paths[0] == len(specs)


@pytest.mark.parametrize("gen", [2, 3])
def test_random_jobs_default_params(gen):
    specs = [synth_wfa_job(seed, ref_len=3000 + 37 * seed, n_vars=6 + seed % 9, noise=0.003 + 0.001 * (seed % 5))[0]
             for seed in range(1, 49)]
    paths = compare(specs, 500, 500, gen=gen)
    assert paths[0] >= 40, paths


@pytest.mark.parametrize("gen", [2, 3])
def test_random_jobs_wide_parameters(gen):
    r = _Rng(77)
    m, d = model(), oracle()
    tot = [0, 0, 0]
    for _ in range(40):
        prune = [0, 20, 100, 500][r.randint(0, 3)]
        max_ed = [8, 60, 150, 500][r.randint(0, 3)]
        specs = []
        for _ in range(6):
            L = [200, 600, 2000, 6000][r.randint(0, 3)]
            specs.append(synth_wfa_job(r.next(), ref_len=max(L, 800), n_vars=r.randint(0, 24), n_homs=r.randint(0, 6),
                                       noise=[0.0, 0.002, 0.01, 0.03][r.randint(0, 3)], multiallelic=0.3)[0])
        p = compare(specs, prune, max_ed, m, d, gen=gen)
        tot = [a + b for a, b in zip(tot, p)]
    assert tot[0] > 60, tot   # the rest outgrew the compact state (no pruning / heavy noise): dense-band kernel
