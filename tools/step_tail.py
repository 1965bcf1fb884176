#!/usr/bin/env python3
"""The last milliseconds of one training step from a rocprofv3 rocpd database (kernel trace): every kernel that STARTS in the final
`ms` before the step's last Adam launch ends, per stream, with start offsets relative to that end.  usage: step_tail.py <db> [ms=2.5]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
namecol = 'name' if 'name' in cols else 'kernel_name'
rows = c.execute('select %s, start, end, stream_id from kernels order by start' % namecol).fetchall()
ends = [r[2] for r in rows if 'adam_kernel' in r[0]][1::2]
t1 = ends[-2]
for n, s, e, q in rows:
    if t1 - ms * 1e6 <= s <= t1 + 0.5e6:
        print('stream %s  %+8.3f ms  %7.1f us  %s' % (q, (s - t1) / 1e6, (e - s) / 1e3, re.sub(r'\(.*', '', n).replace('void ', '').replace('srhip::', '')[:70]))
