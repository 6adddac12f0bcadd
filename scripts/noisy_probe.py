"""C2-noisy (C=60, e=0.15) blocks: where the time goes (heuristic vs main search, pops, pruning)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from hiphase_amd import ResidentBatch, synth_block
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
blocks = [synth_block(n, 60, 20, 0.15, 0.02, 31000 + i)[0] for i in range(nb)]
rb = ResidentBatch(blocks); rb.solve(); ms = rb.solve(); res, ctrs, _ = rb.results()
mc = np.array([c.reserved[1] for c in ctrs], float) / 2.4e6; hc = np.array([c.reserved[0] for c in ctrs], float) / 2.4e6
mp = np.array([c.main_pops for c in ctrs], float); sp = np.array([c.sub_pops for c in ctrs], float)
pr = np.array([r.statistics.pruned_solutions for r in res], float)
print(f"blocks={nb} N={n} kernel_ms={ms:.1f} -> {nb*n/ms/1e3:.2f} M hets/s")
print(f"heuristic ms mean {hc.mean():.1f} max {hc.max():.1f}; main ms mean {mc.mean():.1f} max {mc.max():.1f}")
print(f"sub_pops/het {sp.mean()/n:.1f}  main_pops/het {mp.mean()/n:.1f}  pruned/het {pr.mean()/n:.2f}  us per main pop {1e3*mc.mean()/mp.mean():.2f}")
