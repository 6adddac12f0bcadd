"""One noisy block through the resident solver (for rocprofv3 --kernel-trace): which launches its time is made of."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from hiphase_amd import ResidentBatch, synth_block
n, c, e = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (2500, 60, 0.15)
blk, _ = synth_block(n, c, 20, e, 0.02, 4242)
rb = ResidentBatch([blk])
rb.solve()
print('kernel_ms', rb.solve())
rb.close()
