"""Print (calls, avg us, total ms) per kernel from a rocprofv3 --kernel-trace --stats output directory (csv)."""
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
        name = re.sub(r'\(.*', '', r['Name'].replace('void ', ''))[:80]
        print('%-82s calls %6s avg %8.1f us total %8.2f ms' % (name, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
