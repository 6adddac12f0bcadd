"""Sums the counters the W3_STATS build's sampled workgroups print (hp_wfa3_kernel.hip): steps, tiles and lane use per read."""
import re, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(int))
for l in open(sys.argv[1]):
    m = re.match(r'w3 G(\d+) W(\d+) wg \d+: (.*)', l)
    if not m: continue
    key = (int(m.group(1)), int(m.group(2)))
    for k, v in re.findall(r'([a-z\-]+) (\d+)', m.group(3)):
        tot[key][k] += int(v)
    tot[key]['wgs'] += 1
for key, d in sorted(tot.items()):
    # (counters kept inside divergent code - jobs, rounds, control passes, inserts and what follows - only count the steps in which
    # lane 0's group took part: its own group's events, roughly one eighth / one quarter of the workgroup's)
    print(f"G{key[0]} W{key[1]} group 0 of each sampled workgroup: rounds {d['rounds']}, jobs {d['jobs']}, finished waves {d['finished-waves']}, inserts {d['inserts']} of which not appended {d['slow-inserts']} "
          f"(scan iterations {d['scan-iters']}, shift iterations {d['shift-iters']}), build chunks {d['build-chunks']}")
    j = max(1, d['jobs']); G = key[0]
    print(f"G{key[0]} W{key[1]}: {d['wgs']} workgroups sampled, {d['jobs']} jobs; per job: rounds {d['rounds']/j:.1f}, group tiles {d['group-tiles']/j:.1f} ({d['group-tiles']/max(1,d['rounds']):.2f} per round), "
          f"targets {d['act']/j:.0f} ({d['act']/max(1,d['group-tiles']):.2f} of {G} lanes per tile), with a wave {d['has']/j:.0f}, committed {d['committed']/j:.0f}, discarded {d['discarded-lanes']/j:.1f}, "
          f"long extensions {d['long-ext-lanes']/j:.0f}, inserts {d['inserts']/j:.1f}, build chunks {d['build-chunks']/j:.1f}; wave steps per job-slot {d['wave-steps']*(64//G)/j:.0f}, control passes per wave step {d['control-passes']/max(1,d['wave-steps']):.2f}")
