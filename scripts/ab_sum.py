import json, sys
for f in sys.argv[1:]:
    rows = []
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        rows.append((d['ms_per_step'], d['roofline']['kernel_ms'], d['resident']['ms_per_step']))
    print(f, ' | '.join(f"{a:.1f} / {b:.1f} / {c:.1f}" for a, b, c in rows), '  (streamed ms per step / graph-WFA span / resident ms per step)')
