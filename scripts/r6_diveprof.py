"""A dive step of the main search by part (library built with -DHP_MAIN_PROF=2, HP_LIB pointing at it): clean data, every pop a dive step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from hiphase_amd import ResidentBatch, synth_block
for n, c, e in ((5000, 30, 0.01), (4000, 60, 0.01)):
    blk, _ = synth_block(n, c, 20, e, 0.02, 4242)
    rb = ResidentBatch([blk]); rb.solve(); ms = rb.solve()
    ctr = rb.results()[1][0]
    v = [ctr.sub_pops, ctr.evals, ctr.cells, ctr.nodes_created] + list(ctr.reserved)
    pops = ctr.main_pops
    names = ['head (rings, tracker remove)', 'expand', 'keys + best of four', 'tracker add + record store', 'push3', 'prune check', 'take the child (Cur, fast state)']
    print(n, c, e, 'kernel_ms', round(ms, 2), 'pops', pops, 'ticks per pop:', ', '.join('%s %d' % (a, b / max(pops, 1)) for a, b in zip(names, v)), '| sum', sum(v) // max(pops, 1))
    rb.close()
