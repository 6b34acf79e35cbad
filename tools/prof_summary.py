#!/usr/bin/env python3
"""Turn a rocprofv3 results database (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into the
per-kernel markdown summary committed under profiles/.   usage: prof_summary.py <results.db> <steps>"""
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    print('| kernel | calls/step | us/step | avg us/launch | % GPU time |')
    print('|---|---:|---:|---:|---:|')
    tot = 0.0
    for name, calls, dur, avg, pct in rows:
        name = name.replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0]
        print(f'| `{name}` | {calls / steps:.1f} | {dur / steps:.1f} | {avg:.1f} | {pct:.1f} |')
        tot += dur / steps
    print(f'\nGPU kernel time per step: {tot:.1f} us')


if __name__ == '__main__':
    main()
