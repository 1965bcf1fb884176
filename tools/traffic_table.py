"""Per-kernel HBM bytes of a profiled run (tools/gpu_r5_traffic.sh): calls per step, average MB read (FETCH_SIZE, KiB units, no
correction applied: 16-byte-per-lane streams under-count by up to 2x on gfx950 -- MI355X_MICROARCH.md) and written (WRITE_SIZE), MB per step.
With a kernel-trace run of the same command under <dir>/kt (rocprofv3 --kernel-trace --stats) two more columns: the kernel's average duration
INSIDE the step (three streams share the chip) and the HBM rate that goes with it, (read + written) / duration.
usage: traffic_table.py <dir> <steps profiled> [rows]"""
import csv, glob, os, re, sys, collections
root, steps = sys.argv[1], float(sys.argv[2])


def short(n):
    return re.sub(r'\(.*', '', n.replace('void ', ''))[:78]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('fetch', 'write'):
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob(os.path.join(root, 'kt', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r['Name'])] = float(r['AverageNs']) / 1e3
rows = []
for k, c in agg.items():
    n = max(len(c.get('FETCH_SIZE', [])), len(c.get('WRITE_SIZE', [])))
    rd = sum(c.get('FETCH_SIZE', [])) * 1024 / 1e6
    wr = sum(c.get('WRITE_SIZE', [])) * 1024 / 1e6
    rows.append((rd + wr, k, n, rd, wr))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total %.1f GB per step (read %.1f, written %.1f)' % (tot / steps / 1e3, sum(r[3] for r in rows) / steps / 1e3, sum(r[4] for r in rows) / steps / 1e3))
print('%-80s %8s %10s %10s %10s %10s %8s' % ('kernel', 'calls/st', 'read MB', 'write MB', 'MB/step', 'us in-step', 'TB/s'))
for t, k, n, rd, wr in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    us = dur.get(k)
    rate = ((rd + wr) / max(n, 1)) / us if us else None          # MB / us = TB/s
    print('%-80s %8.1f %10.1f %10.1f %10.1f %10s %8s' % (k, n / steps, rd / max(n, 1), wr / max(n, 1), t / steps,
                                                       '%.1f' % us if us else '-', '%.2f' % rate if rate else '-'))
