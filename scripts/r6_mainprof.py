"""Main-search time by phase (library built with -DHP_MAIN_PROF=1, HP_LIB pointing at it)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from hiphase_amd import ResidentBatch, synth_block
for n, c, s, e in ((2500, 60, 20, 0.15), (4000, 60, 20, 0.15), (2500, 30, 20, 0.15), (5000, 30, 20, 0.02)):
    blk, _ = synth_block(n, c, s, e, 0.02, 4242)
    rb = ResidentBatch([blk])
    rb.solve()
    ms = rb.solve()
    ctr = rb.results()[1][0]
    r = list(ctr.reserved)
    main = r[1]
    exp, store = (r[0] & 0xFFFFFFFF) << 10, (r[0] >> 32) << 10
    pop, fam = (r[2] & 0xFFFFFFFF) << 10, (r[2] >> 32) << 10
    jumps = ctr.sub_pops
    mp = ctr.main_pops
    f = lambda x: '%.1f M (%.0f %%)' % (x / 1e6, 100.0 * x / max(main, 1))
    print(n, c, e, 'kernel_ms', round(ms, 1), 'main_ms (100 MHz clock)', round(ctr.cells / 1e5, 2), 'ticks per us', round(main / max(ctr.cells, 1) * 100), 'main_pops', mp, 'jumps', jumps, 'main ticks', round(main / 1e6, 1), 'M = ', round(main / max(mp, 1)), 'per pop |',
          'expand', f(exp), 'store+push3', f(store), 'push+pop', f(pop), 'fam', f(fam), 'rest', f(main - exp - store - pop - fam),
          '| per jump: pop', round(pop / max(jumps, 1)), 'fam', round(fam / max(jumps, 1)), '| per pop: expand', round(exp / max(mp, 1)), 'store', round(store / max(mp, 1)))
    rb.close()
