#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Values are KiB per dispatch; FETCH_SIZE is
doubled for kernels whose reads are wide coalesced streams (the gfx950 tally counts 128-B requests at
64 B).   usage: pmc_summary.py <fetch.db> <write.db>"""
import collections
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = collections.defaultdict(list)
    for name, val in cur.execute(
            'select kernel_name, value from counters_collection where counter_name = ?', (counter,)):
        name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name].append(val)
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    print('| kernel | launches | FETCH_SIZE MiB/launch (raw) | x2 (wide-stream correction) | WRITE_SIZE MiB/launch |')
    print('|---|---:|---:|---:|---:|')
    rows = []
    for k in fetch:
        f = sum(fetch[k]) / len(fetch[k]) / 1024.0
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) / 1024.0
        rows.append((f + w, k, len(fetch[k]), f, w))
    for _, k, n, f, w in sorted(rows, reverse=True)[:24]:
        print(f'| `{k}` | {n} | {f:.1f} | {2 * f:.1f} | {w:.1f} |')


if __name__ == '__main__':
    main()
