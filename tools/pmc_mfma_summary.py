#!/usr/bin/env python3
"""MFMA-pipe / LDS counters per kernel from the two passes of tools/profile_mfma.sh (markdown on stdout).
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); executed GFLOP = MOPS_F32 x 512."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out')


def load(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
        by.setdefault((name, int(r['Grid_Size']) // 256), collections.defaultdict(list))[r['Counter_Name']].append(
            float(r['Counter_Value']))
    return {k: {n: sum(x) / len(x) for n, x in v.items()} | {'_n': len(next(iter(v.values())))} for k, v in by.items()}


m = load(os.path.join(G, 'pmc_mfma', 'm_counter_collection.csv'))
lp = os.path.join(G, 'pmc_lds', 'l_counter_collection.csv')
l = load(lp) if os.path.exists(lp) else {}          # (the LDS pass is optional)
has16 = any('SQ_INSTS_VALU_MFMA_MOPS_BF16' in v for v in m.values())
print('| kernel | blocks | launches | GPU cycles | MFMA pipe busy | executed GFLOP (fp32 MFMA)' + (' | executed GFLOP (bf16 MFMA)' if has16 else '')
      + ' | LDS bank-conflict cycles / LDS active cycles |')
print('|---|---:|---:|---:|---:|---:|---:|' + ('---:|' if has16 else ''))
rows = []
for k, v in m.items():
    if not v.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        continue
    cyc = v['GRBM_GUI_ACTIVE'] / 8.0
    busy = 100.0 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)
    lv = l.get(k, {})
    conf = lv.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(lv.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0)
    rows.append((cyc * v['_n'], f"| `{k[0]}` | {k[1]} | {v['_n']} | {cyc:,.0f} | {busy:.1f} % | "
                 f"{v.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512 / 1e9:.2f} | "
                 + (f"{v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512 / 1e9:.2f} | " if has16 else '') + f"{conf:.3f} |"))
for _, line in sorted(rows, reverse=True):
    print(line)
