"""CPU model of the pull-based band formulation used by hp_wfa_kernel.hip (debug aid): same candidate rules,
same tie rule, same skip/emit rules, dictionaries instead of the dense band."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiphase_amd import _ffi
from hiphase_amd.wfa_graph import make_jobs
from oracle_ffi import oracle, OracleGraph

NONE, INT, INTR, ENDL = 0, 1, 3, 4

def model(g, read, prune=2**64-1, max_ed=1000):
    n_nodes = g.num_nodes()
    seqs = [g.node_seq(i) for i in range(n_nodes)]
    children = [g.node_edges(i) for i in range(n_nodes)]
    last = n_nodes - 1
    L = len(read)
    def run(seq, o, pos, maxlen):
        k = 0
        while k < maxlen and seq[o + k] == read[pos + k]: k += 1
        return k
    prev = {}   # (n,d) -> (mo, kind, set)
    maxfront = {}
    inj = {(0, 0): [frozenset([0])]}
    farthest = 0; min_prog = 0
    for ed in range(0, max_ed + 1):
        cur = {}
        finals = []
        for n in range(n_nodes):
            diags = set()
            for (nn, d) in prev:
                if nn == n: diags.update([d - 1, d, d + 1])
            for (nn, d) in inj:
                if nn == n: diags.add(d)
            seq = seqs[n]; ln = len(seq)
            for d in sorted(diags):
                cands = []
                r = prev.get((n, d + 1));
                if r and (r[1] & 1): cands.append((r[0] + 1, r[2]))
                r = prev.get((n, d));
                if r and r[1] == INTR: cands.append((r[0] + 1, r[2]))
                r = prev.get((n, d - 1));
                if r and r[1] in (INTR, ENDL): cands.append((r[0], r[2]))
                for s in inj.pop((n, d), []): cands.append((0, s))
                if not cands: continue
                omax = max(o for o, _ in cands)
                pos0 = d + omax
                room = min(ln - omax, L - pos0) if 0 <= pos0 <= L else 0
                E = omax + run(seq, omax, pos0, max(room, 0))
                best = set()
                for o, s in cands:
                    if o == omax or run(seq, o, d + o, omax - o) == omax - o: best |= s
                best = frozenset(best)
                pos_end = d + E
                if n == last and E == ln and pos_end == L: finals.append(best)
                mf = maxfront.get((n, d), 0)
                if E < mf or pos_end < min_prog: continue
                maxfront[(n, d)] = E
                farthest = max(farthest, pos_end)
                if E == ln:
                    if n == last:
                        if pos_end < L: cur[(n, d)] = (E, ENDL, best)
                    else:
                        for c in children[n]:
                            inj.setdefault((c, d + E), []).append(best | {c})
                else:
                    cur[(n, d)] = (E, INTR if pos_end < L else INT, best)
        if finals:
            s = set()
            for f in finals: s |= f
            return 0, ed, sorted(s)
        prev = cur
        if farthest > prune: min_prog = farthest - prune
    return 1, max_ed, []

if __name__ == "__main__":
    from wfa_util import synth_wfa_job
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    spec, _ = synth_wfa_job(seed, ref_len=3000 + 37 * seed, n_vars=6 + seed % 9, noise=0.003 + 0.001 * (seed % 5))
    d = oracle()
    jobs, keep = make_jobs([spec])
    st = C.c_int(0)
    g = OracleGraph(handle=d.hpo_graph_from_job(C.byref(jobs[0]), 500, C.byref(st)))
    o = g.edit_distance(list(spec.read), prune_distance=500)
    m = model(g, spec.read, prune=500, max_ed=500)
    print("oracle", o)
    print("model ", m)
    print("equal", (o[0], o[1], o[2]) == m)
