#!/usr/bin/env python3
"""Timeline view of one training step from a rocprofv3 rocpd database: step boundaries = the Adam launches; per stream the
busy time inside the step, the idle gaps of the busiest (main) stream, and what the other streams ran during the largest gaps."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('srhip::', '')
    return n[:60]


def main(path, which=-2):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute('select %s, start, end, stream_id from kernels order by start' % namecol).fetchall()
    adam = [r for r in rows if 'adam_kernel' in r[0]]
    # two Adam launches per step (G, D): a step = (end of D's Adam of step i-1, end of D's Adam of step i)
    ends = [r[2] for r in adam][1::2]
    t0, t1 = ends[which - 1], ends[which]
    step = [r for r in rows if r[1] >= t0 and r[2] <= t1 + 1]
    print('step window %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
    streams = {}
    for n, s, e, q in step:
        streams.setdefault(q, []).append((s, e, n))
    main_q = max(streams, key=lambda q: sum(e - s for s, e, _ in streams[q]))
    for q, ks in sorted(streams.items()):
        print('stream %s: %5d kernels, busy %.2f ms, first start +%.2f ms, last end +%.2f ms%s' % (
            q, len(ks), sum(e - s for s, e, _ in ks) / 1e6, (ks[0][0] - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6, '  <- main' if q == main_q else ''))
    ks = streams[main_q]
    gaps = []
    prev_end, prev_name = t0, '(step start)'
    for s, e, n in ks:
        if s - prev_end > 0:
            gaps.append((s - prev_end, prev_end, s, prev_name, n))
        if e > prev_end:
            prev_end, prev_name = e, n
    tot = sum(g[0] for g in gaps)
    print('main stream idle inside the step: %.2f ms in %d gaps; > 20 us: %.2f ms in %d gaps; > 100 us: %.2f ms in %d gaps' % (
        tot / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 20e3) / 1e6, sum(1 for g in gaps if g[0] > 20e3),
        sum(g[0] for g in gaps if g[0] > 100e3) / 1e6, sum(1 for g in gaps if g[0] > 100e3)))
    for d, a, b, pn, nn in sorted(gaps, reverse=True)[:14]:
        others = {}
        for q, oks in streams.items():
            if q == main_q:
                continue
            busy = sum(min(e, b) - max(s, a) for s, e, _ in oks if e > a and s < b)
            others[q] = busy / d
        print('  gap %7.1f us at +%6.2f ms  after %-45s before %-45s other streams busy %s' % (
            d / 1e3, (a - t0) / 1e6, short(pn), short(nn), {q: round(v, 2) for q, v in others.items()}))
    # the tail of the step: everything after the last weight-gradient kernel of the generator's backward is the discriminator's backward
    import os
    tail_ms = float(os.environ.get('TAIL_MS', '16'))
    w0 = t1 - int(tail_ms * 1e6)
    agg = {}
    busy = {}
    for n, s_, e_, q in step:
        if e_ <= w0:
            continue
        d = e_ - max(s_, w0)
        a = agg.setdefault((q, short(n)), [0, 0])
        a[0] += 1
        a[1] += d
        busy[q] = busy.get(q, 0) + d
    print('last %.0f ms of the step: busy per stream %s' % (tail_ms, {q: round(v / 1e6, 2) for q, v in sorted(busy.items())}))
    for (q, n), (cnt, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print('   stream %s  %-62s calls %4d  %7.2f ms  avg %7.1f us' % (q, n, cnt, d / 1e6, d / cnt / 1e3))
    # concurrency histogram
    ev = []
    for n, s, e, q in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur, last, hist = 0, t0, {}
    for t, dlt in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        last = t
        cur += dlt
    print('time with k kernels in flight:', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
