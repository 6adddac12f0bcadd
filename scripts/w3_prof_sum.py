"""Sums the s_memtime phase timers the W3_PROF build's sampled workgroups print (hp_wfa3_kernel.hip): share of time per phase of the lockstep step."""
import re, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(int))
for l in open(sys.argv[1]):
    m = re.match(r'w3prof G(\d+) W(\d+) wg \d+: (.*)', l)
    if not m: continue
    d = {k: int(v) for k, v in re.findall(r'([a-z\-]+) (\d+)', m.group(3))}
    if any(v > 1 << 40 for v in d.values()): continue   # (a timer that wrapped)
    key = (int(m.group(1)), int(m.group(2)))
    for k, v in d.items(): tot[key][k] += v
    tot[key]['wgs'] += 1
for key, d in sorted(tot.items()):
    T = max(1, d['total']); st = max(1, d['steps'])
    print(f"G{key[0]} W{key[1]}: {d['wgs']} workgroups, {st} wave steps, {T / st:.0f} ticks per step")
    for k in ('between', 'control', 'candidates', 'issue', 'extension', 'ties', 'capped-decide', 'commit-write', 'capins', 'finals', 'pre-inject', 'inject'):
        print(f"   {k:14s} {100.0 * d.get(k, 0) / T:6.2f} %   {d.get(k, 0) / st:8.0f} ticks/step")
