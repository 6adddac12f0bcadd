import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from hiphase_amd import ResidentBatch, synth_block
for n, c, s, e in ((2500, 60, 20, 0.15), (4000, 60, 20, 0.15), (2500, 30, 20, 0.15)):
    blk, _ = synth_block(n, c, s, e, 0.02, 4242)
    for env in ({}, {"HP_NO_SEGMENTS": "1"}):
        for k in ("HP_NO_SEGMENTS",):
            os.environ.pop(k, None)
        os.environ.update(env)
        rb = ResidentBatch([blk])
        rb.solve()
        t = time.perf_counter(); ms = rb.solve(); dt = (time.perf_counter() - t) * 1e3
        out = rb.results()
        res, ctr = out[0], out[1]
        print(n, c, e, env, 'kernel_ms', round(ms, 1), 'wall', round(dt, 1), 'stats', res[0].statistics.as_tuple() if hasattr(res[0], 'statistics') else res[0], 'sub_pops, main_pops, evals, cells, nodes', ctr[0].as_tuple(), 'cycles heuristic / main (shader clock)', list(ctr[0].reserved)[:2])
        rb.close()
