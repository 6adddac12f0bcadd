"""Wall-clock cost of the one-block-per-call entry point hp_astar_solve (create + pack + upload + solve + download +
destroy) for small blocks, where fixed overhead dominates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hiphase_amd import astar_solver, synth_block
for n in (15, 200, 2000):
    blk = synth_block(n, 30, 20, 0.01, 0.02, 99 + n)[0]
    astar_solver(0, blk)
    t0 = time.perf_counter()
    k = 30
    for _ in range(k):
        astar_solver(0, blk)
    print(f"N={n:5d}: {(time.perf_counter() - t0) / k * 1e3:8.2f} ms per hp_astar_solve call", flush=True)
