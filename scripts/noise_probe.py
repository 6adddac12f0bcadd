"""Whole-path timing on read-bearing blocks with more read noise than the bench's 0.3 % (where most reads outgrow the
compact graph-WFA kernels and take the dense-band pass): looks for pathologies, not for speed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hiphase_amd.blocks import BlockSet
from hiphase_amd.read_parsing import GlobalRealignmentConfig
from hiphase_amd.synth_reads import synth_read_block

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
specs = [synth_read_block(900 + i, 60 + 40 * (i % 7), block_index=i, noise=noise)[0] for i in range(n_blocks)]
reads = sum(len(s.records) for s in specs)
bs = BlockSet(specs, config=GlobalRealignmentConfig(max_edit_distance=4000))   # (no CIGAR views here: keep every read global)
for _ in range(2):
    bs.solve()
t0 = time.perf_counter()
st = [bs.solve() for _ in range(3)]
dt = (time.perf_counter() - t0) / 3
res = bs.results()
print(f"noise {noise}: {n_blocks} blocks, {reads} reads, {dt * 1e3:.1f} ms per solve, stages {[round(x, 1) for x in st[-1]]}, "
      f"local fallbacks {sum(r.local_aligned for r in res)}, global {sum(r.global_aligned for r in res)}")
bs.close()
