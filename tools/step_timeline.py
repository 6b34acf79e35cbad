#!/usr/bin/env python3
"""Timeline of one graph-replayed step from a rocprofv3 --kernel-trace database: per-kernel start / duration
and the idle gap before it, plus wall / busy / idle totals.   usage: step_timeline.py <results.db> [step]"""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 15
c = sqlite3.connect(db).cursor()
rows = list(c.execute('select name,start,end,queue_id from kernels order by start'))
short = lambda n: re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0][:58]
idx = [i for i, r in enumerate(rows) if 'phase_embed_fwd' in r[0] or 'phase_embed_begin' in r[0]]
for s in (which - 1, which, which + 1):
    seg = rows[idx[s]:idx[s + 1]]
    iv = sorted((r[1], r[2]) for r in seg)
    busy, gaps, (cs, ce) = 0, 0, iv[0]
    for a, b in iv[1:]:
        if a <= ce:
            ce = max(ce, b)
        else:
            busy += ce - cs; gaps += a - ce; cs, ce = a, b
    busy += ce - cs
    print('step %d: %d kernels, start-to-start %.1f us, busy (union) %.1f, idle %.1f, sum of durations %.1f' % (
        s, len(seg), (rows[idx[s + 1]][1] - seg[0][1]) / 1e3, busy / 1e3, gaps / 1e3,
        sum(r[2] - r[1] for r in seg) / 1e3))
a, b = idx[which], idx[which + 1]
prev_end = None
for r in rows[a:b + 1]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0
    print('%-58s q%-2s start %8.1f dur %7.1f gap %6.1f' % (short(r[0]), r[3], (r[1] - rows[a][1]) / 1e3,
                                                          (r[2] - r[1]) / 1e3, gap))
    prev_end = max(prev_end or 0, r[2])
