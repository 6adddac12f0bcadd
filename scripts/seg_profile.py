"""Per-segment shader-clock profile of the sub-solver loop (HP_SEG_PROFILE=1)."""
import os, sys
os.environ["HP_SEG_PROFILE"] = "1"
os.environ["HP_NO_SEGMENTS"] = "1"   # the segment kernel carries no instrumentation
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hiphase_amd import ResidentBatch, synth_block
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blocks = [synth_block(5000, 30, 20, 0.01, 0.02, 20250509 + i)[0] for i in range(nb)]
rb = ResidentBatch(blocks); ms = rb.solve(); res, ctrs, _ = rb.results()
names = ["loop head + LDS rings", "row metadata loads", "word loads + scoring", "wave_sum8", "totals + keys", "store + push/pop"]
seg = [0] * 6
for c in ctrs:
    for k in range(3):
        v = c.reserved[k]
        seg[2 * k] += (v & 0xFFFFFFFF) << 10; seg[2 * k + 1] += (v >> 32) << 10
pops = sum(c.sub_pops for c in ctrs)
print(f"blocks={nb} kernel_ms={ms:.1f} sub_pops={pops}")
for n, s in zip(names, seg):
    print(f"  {n:26s} {s / pops:8.0f} cycles/pop")
print(f"  {'total':26s} {sum(seg) / pops:8.0f} cycles/pop")
