#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one row per kernel x counter)."""
import csv, sys, collections, re
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*', '', n); return n[:70]
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('##', path)
    for k, cs in agg.items():
        if not k.startswith('srhip::fast') : continue
        print(k)
        for c, v in sorted(cs.items()):
            print('    %-34s n=%-4d avg=%.4g' % (c, len(v), sum(v) / len(v)))
