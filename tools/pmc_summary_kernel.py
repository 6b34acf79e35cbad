#!/usr/bin/env python3
"""Per-kernel PMC ratios from a rocprofv3 --pmc counter_collection.csv: usage pmc_summary_kernel.py <csv> [substr]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
by = collections.OrderedDict()
for r in rows:
    if sub not in r['Kernel_Name']:
        continue
    by.setdefault((r['Kernel_Name'][:90], r['Grid_Size']), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for (k, grid), v in by.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    wc = m.get('SQ_WAVE_CYCLES')
    print(k, 'grid', grid, 'launches', len(next(iter(v.values()))))
    for n, x in sorted(m.items()):
        print(f'    {n:28s} {x:14.0f}' + (f'  /wave_cycles = {x / wc:.3f}' if wc else ''))
