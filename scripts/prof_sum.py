"""Sums the counters the W2_PROF build's sampled workgroups print (scripts/prof_wfa2.sh): share of time per phase of the lockstep step."""
import re, sys
tot = {}; n = 0
for l in open(sys.argv[1]):
    if not l.startswith('wg '): continue
    d = {k: int(v) for k, v in re.findall(r'([a-zA-Z>+\-]+[a-zA-Z0-9>+\-]*) (\d+)', l)}
    if any(v > 1 << 40 for v in d.values()): continue
    n += 1
    for k, v in d.items(): tot[k] = tot.get(k, 0) + v
T = tot['total']; it = max(1, tot.get('iters', 1))
print(n, 'workgroups')
for k, v in tot.items():
    if k in ('wg',): continue
    if k in ('iters', 'ctlpasses', 'tile-lanes', 'has-lanes', 'ext2', 'tieslow', 'hashprobe', 'capins'): print(f"{k:14s} per step {v / it:.3f}")
    else: print(f"{k:14s} {100 * v / T:6.2f} %   {v / it:8.0f} ticks/step")
