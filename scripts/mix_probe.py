"""Times the heavy-tailed WGS-like block mix and a single C2 block (latency-bound cases); honours HP_NO_SEGMENTS /
HP_SEG_TARGET / HP_NO_CTAB from the environment."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hiphase_amd.astar_phaser import ResidentBatch

def run(tag, a):
    blocks = bench.make_blocks(a, 0)
    rb = ResidentBatch(blocks)
    rb.solve()
    ms = min(rb.solve() for _ in range(3))
    rb.close()
    print(f"{tag} blocks={len(blocks)} ms={ms:.1f}", flush=True)

base = dict(replay=None, hets=5000, coverage=30, span=20, error=0.01)
run("wgs12000", argparse.Namespace(workload="wgs", blocks=12000, **base))
run("c2x1", argparse.Namespace(workload="c2", blocks=1, **base))


