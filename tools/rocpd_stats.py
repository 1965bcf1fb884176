#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(name, calls, total ms, avg us, %), the same content as rocprofv3's kernel_stats.csv."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '')
    return name[:150]


def main(path, top=45):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute('select %s, start, end from kernels' % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    for col in ('queue_id', 'stream_id'):                   # per hardware queue / stream: busy time and span
        if col in cols:
            per = {}
            for q, s_, e_ in c.execute('select %s, start, end from kernels' % col).fetchall():
                a = per.setdefault(q, [0, 0, None, None])
                a[0] += 1
                a[1] += e_ - s_
                a[2] = s_ if a[2] is None else min(a[2], s_)
                a[3] = e_ if a[3] is None else max(a[3], e_)
            for q, (cnt, busy, s0, e0) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
                print('# %s %s: %d kernels, busy %.1f ms over a span of %.1f ms' % (col, q, cnt, busy / 1e6, (e0 - s0) / 1e6))
    print('# kernels: %d dispatches, %.3f ms total GPU kernel time' % (len(rows), tot / 1e6))
    print('%-9s %-11s %-10s %-6s %s' % ('calls', 'total_ms', 'avg_us', 'pct', 'kernel'))
    for n, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-9d %-11.3f %-10.2f %-6.2f %s' % (cnt, t / 1e6, t / cnt / 1e3, 100.0 * t / tot, short(n)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
