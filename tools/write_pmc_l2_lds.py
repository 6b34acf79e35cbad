"""gpurun_out/r06p/pmc_l2, pmc_ldsx (tools/profile_r06_gemm_xp.sh) -> profiles/r06_pmc_l2_lds.md: L2 and LDS counters per kernel and launch."""
import collections
import csv
import subprocess

head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()


def load(tag):
    rows = list(csv.DictReader(open(f'gpurun_out/r06p/{tag}/l_counter_collection.csv')))
    agg, disp, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set), collections.defaultdict(float)
    for r in rows:
        k = (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70], int(r['Grid_Size']) // int(r['Workgroup_Size']))
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in disp[k]:
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        disp[k].add(r['Dispatch_Id'])
    return {k: ({c: v / len(disp[k]) for c, v in a.items()}, len(disp[k]), dur[k] / len(disp[k])) for k, a in agg.items()}


l2, lds = load('pmc_l2'), load('pmc_ldsx')
keys = sorted(l2, key=lambda k: -l2[k][0].get('TCC_REQ_sum', 0) * l2[k][1])[:16]
out = [f'# Round 6 (commit {head}) -- L2 and LDS counters of the headline step\'s kernels, per launch',
       '',
       'Commands: `tools/profile_r06_gemm_xp.sh` (two separate `rocprofv3 --kernel-trace --pmc` passes of `bench.py`, eager launches, 8 x 300).  '
       'L2 hit rate = TCC_HIT_sum / TCC_REQ_sum; L1 -> L2 reads = TCP_TCC_READ_REQ_sum (64-byte requests); LDS conflict share = '
       'SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles of all LDS cycles).',
       '',
       '| kernel | workgroups | launches | us / launch (eager, under the counters) | L2 requests | L2 hit rate | L1 -> L2 read requests | LDS instructions | LDS-active cycles | conflict share |',
       '|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|']
for k in keys:
    a, n, us = l2[k]
    b = lds.get(k, ({}, 0, 0))[0]
    act = b.get('SQ_LDS_IDX_ACTIVE', 0)
    out.append(f'| `{k[0]}` | {k[1]} | {n} | {us:.1f} | {a.get("TCC_REQ_sum", 0):,.0f} | {100 * a.get("TCC_HIT_sum", 0) / max(1, a.get("TCC_REQ_sum", 1)):.1f} % | '
               f'{a.get("TCP_TCC_READ_REQ_sum", 0):,.0f} | {b.get("SQ_INSTS_LDS", 0):,.0f} | {act:,.0f} | '
               f'{(100 * b.get("SQ_LDS_BANK_CONFLICT", 0) / act if act else 0):.1f} % |')
out += ['',
        'Reading: the mesh kernel pulls 25.8 M 64-byte reads per launch through L1 (1.65 GB of L2 -> L1 traffic + the hits\' share of its 2.8 GB operand stream) at a 90 % L2 hit '
        'rate -- the blend shapes are L2-resident and streamed per 16-sample workgroup; its LDS conflicts (49 %) cost no time (`r06_experiments.md` section 5).  '
        'The 2401-row `gemm_xp` launches miss L2 on 35 - 60 % of their requests: their operands come from the launch before through the Infinity Cache, once per XCD.']
open('profiles/r06_pmc_l2_lds.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:12]))
