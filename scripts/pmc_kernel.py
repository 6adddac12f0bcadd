"""Sum a rocprofv3 --pmc counter csv per kernel name: python scripts/pmc_kernel.py <counter_collection.csv> [name filter]"""
import csv, sys, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ''
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    if flt and flt not in r['Kernel_Name']: continue
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in tot:
    print(k, {c: (round(v), n[(k, c)]) for c, v in tot[k].items()})
