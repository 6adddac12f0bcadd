"""Single-block latency and heavy-tailed batch: segment-parallel heuristic on vs off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from hiphase_amd import ResidentBatch, synth_block

def run(blocks, label):
    for mode in ("seq", "seg"):
        if mode == "seq": os.environ["HP_NO_SEGMENTS"] = "1"
        else: os.environ.pop("HP_NO_SEGMENTS", None)
        rb = ResidentBatch(blocks); rb.solve(); ms = min(rb.solve() for _ in range(2)); rb.close()
        hets = sum(b.n_variants for b in blocks)
        print(f"{label:28s} {mode}: {ms:9.2f} ms  {hets / ms * 1e3 / 1e6:8.3f} M hets/s", flush=True)

run([synth_block(5000, 30, 20, 0.01, 0.02, 20250509)[0]], "1 x C2 block (5000 hets)")
run([synth_block(5000, 60, 20, 0.15, 0.02, 5)[0]], "1 x C2-noisy (C=60,e=.15)")
rng = np.random.default_rng(12345)
sizes = np.clip(np.exp(rng.normal(np.log(15.0), 2.2, 12000)).astype(int), 2, 4000)
run([synth_block(int(n), 30, 20, 0.01, 0.02, 777 + i)[0] for i, n in enumerate(sizes)], "WGS-like 12000 blocks")
run([synth_block(5000, 30, 20, 0.01, 0.02, 100 + i)[0] for i in range(64)], "64 x C2 blocks")
