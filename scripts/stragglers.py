"""Lists the blocks of the default bench batch with the longest main search (cycles, pops, pruned solutions)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from hiphase_amd import ResidentBatch, synth_block
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blocks = [synth_block(5000, 30, 20, 0.01, 0.02, 20250509 + i)[0] for i in range(nb)]
rb = ResidentBatch(blocks); ms = rb.solve(); res, ctrs, _ = rb.results()
mc = np.array([c.reserved[1] for c in ctrs], float); hc = np.array([c.reserved[0] for c in ctrs], float)
print(f"blocks={nb} kernel_ms={ms:.1f}; heuristic ms mean {hc.mean()/2.4e6:.1f} max {hc.max()/2.4e6:.1f}; main ms mean {mc.mean()/2.4e6:.1f} median {np.median(mc)/2.4e6:.1f} max {mc.max()/2.4e6:.1f}")
for i in np.argsort(-mc)[:8]:
    c, st = ctrs[i], res[i].statistics
    print(f"  block {i}: main {mc[i]/2.4e6:8.1f} ms  heur {hc[i]/2.4e6:7.1f} ms  main_pops {c.main_pops:8d}  pruned {st.pruned_solutions:7d}  est {st.estimated_cost} actual {st.actual_cost}  cycles/main_pop {mc[i]/max(c.main_pops,1):.0f}")
tot = (mc + hc) / 2.4e6
print("per-block total ms percentiles (50/90/99/max):", np.percentile(tot, [50, 90, 99, 100]).round(1), " sum/slots:", round(tot.sum() / min(nb, 6144), 1))
