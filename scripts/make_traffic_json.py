#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of scripts/prof_path.sh (PMC=1) into profiles/round3-style summaries:
  <out>/pmc_summary.txt   per-kernel counter totals, the derived ratios DESIGN.md quotes
  <out>/traffic.json      HBM bytes per read (hp_wfa2_kernel, all three class instantiations) and per het (hp_astar_kernel),
                          keyed by the sha256 of the libhiphase_gpu.so they were measured on (bench.py only reports
                          `roofline.traffic` when that hash matches the library it runs).
FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3); FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for
gfx950 (it reports half of the bytes of a wide read); WRITE_SIZE is taken as reported (uncalibrated)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][-1])
launches = bench["steps"] + bench["warmup"]
reads = bench["kernels"][0]["reads"]
hets = bench["config"]["hets_per_step_per_gpu"]

def group(name):
    if "hp_wfa3_kernel" in name:
        return "hp_wfa3_kernel"
    if "hp_wfa2_kernel" in name:
        return "hp_wfa2_kernel"
    if "hp_astar_kernel" in name:
        return "hp_astar_kernel"
    if "hp_heur_seg_kernel" in name:
        return "hp_heur_seg_kernel"
    if "hp_wfa2_build_kernel" in name:
        return "hp_wfa2_build_kernel"
    return None

# per exact kernel name and counter: median over its dispatches x number of dispatches (one dispatch of a profiled run can be
# an outlier: the passes with counters serialise the kernels, and a kernel that waits for another one then waits in vain)
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        g = group(row.get("Kernel_Name", ""))
        if g:
            vals[(g, row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
tot = collections.defaultdict(float)
disp = collections.Counter()
for (g, _name, c), v in vals.items():
    v = sorted(v)
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    tot[(g, c)] += med * len(v)
    disp[(g, c)] += len(v)

lines = [f"workload: {bench['config']['workload']}", f"launches per pass: {launches} (warm-up {bench['warmup']} + {bench['steps']} steps), {reads} reads and {hets} hets per step", ""]
for k in sorted(tot):
    lines.append(f"{k[0]:22s} {k[1]:20s} total={tot[k]:.6g} dispatches={disp[k]} per_step={tot[k] / launches:.6g}")
lines.append("")
traffic = {"_comment": __doc__.strip().split("\n")[0] + " FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes; per step = total / launches.",
           "workload": bench["config"]["workload"], "launches": launches}
sha = hashlib.sha256(open(os.path.join(root, "hiphase_amd", "libhiphase_gpu.so"), "rb").read()).hexdigest()
def _fatbin_sha(path):   # the library's device code (.hip_fatbin): what the traffic is a property of (bench.py matches on either hash)
    import struct
    b = open(path, "rb").read()
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        name_off, _, _, _, off, size = sh(i)[:6]
        if b[stroff + name_off: b.index(b"\0", stroff + name_off)] == b".hip_fatbin":
            return hashlib.sha256(b[off:off + size]).hexdigest()
    return None
fat = _fatbin_sha(os.path.join(root, "hiphase_amd", "libhiphase_gpu.so"))
for g, unit, n in (("hp_wfa3_kernel", "bytes_per_read", reads), ("hp_wfa2_kernel", "bytes_per_read", reads), ("hp_astar_kernel", "bytes_per_het", hets), ("hp_heur_seg_kernel", "bytes_per_het", hets)):
    if (g, "FETCH_SIZE") in tot and (g, "WRITE_SIZE") in tot:
        b = (2.0 * tot[(g, "FETCH_SIZE")] + tot[(g, "WRITE_SIZE")]) * 1024.0 / launches
        traffic[g] = {unit: b / n, "hbm_bytes_per_step": b, "FETCH_SIZE_KB_per_step": tot[(g, "FETCH_SIZE")] / launches,
                      "WRITE_SIZE_KB_per_step": tot[(g, "WRITE_SIZE")] / launches, "so_sha256": sha, "fatbin_sha256": fat,
                      "source": "profiles/round6/path_pmc_summary.txt"}
        lines.append(f"{g}: HBM traffic (FETCH x 2 + WRITE) = {b / 1e6:.1f} MB per step = {b / n:.0f} {unit.replace('_', ' ')}")
    w, a, wa, wi = (tot.get((g, c)) for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"))
    if w:
        lines.append(f"{g}: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {wa / w:.3f}, SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {a / w:.3f}, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {wi / w:.3f}")
        ins = sum(tot.get((g, c), 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
        lines.append(f"{g}: VALU + SALU + LDS instructions per {unit.split('_')[-1]} = {ins / launches / n:.0f} (VALU {tot.get((g, 'SQ_INSTS_VALU'), 0) / launches / n:.0f}, SALU {tot.get((g, 'SQ_INSTS_SALU'), 0) / launches / n:.0f}, LDS {tot.get((g, 'SQ_INSTS_LDS'), 0) / launches / n:.0f})")
        busy = tot.get((g, "SQ_BUSY_CYCLES"))
        if busy:
            lines.append(f"{g}: SQ_BUSY_CYCLES per step = {busy / launches:.4g}")
    h, m = tot.get((g, "TCC_HIT_sum")), tot.get((g, "TCC_MISS_sum"))
    if h is not None and m is not None and h + m > 0:
        lines.append(f"{g}: L2 hit rate = {h / (h + m):.3f}")
    lines.append("")
open(os.path.join(out, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
