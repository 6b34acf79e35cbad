#!/usr/bin/env python3
"""gpurun_out/r06p (tools/profile_r06.sh) -> profiles/r06_kernel_trace_*.md, r06_pmc_traffic.md, r06_pmc_mfma.md,
r06_bench_lines.md and profiles/traffic.json (per-step HBM-side bytes, per-kernel traffic and MFMA-pipe busy fractions bench.py
quotes, with the commit and the kernel instantiations they were taken on).   usage: write_profiles_r06.py [<commit>]"""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out', 'r06p')
P = os.path.join(ROOT, 'profiles')
head = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True,
                                                              cwd=ROOT).stdout.split()[0]
rd = lambda f: open(os.path.join(G, f)).read()
B = '--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --repeat 1 --minibatch-steps 0'

# ---- kernel traces
for tag, what, cmd in (('c2', 'headline: 8 x 300 full batch, fp32, mesh_blend / mlp_gemm f32_split (default)', 'python3 bench.py --steps 20 --warmup 2'),
                       ('c2f', '8 x 300 full batch, fp32, NEMO_MESH_BLEND=f32 NEMO_MLP_GEMM=f32 (everything on the fp32 MFMA pipe)',
                        'NEMO_MESH_BLEND=f32 NEMO_MLP_GEMM=f32 python3 bench.py --steps 20 --warmup 2'),
                       ('v1', 'one-instance shard: 1 x 300', 'python3 bench.py --instances 1 --steps 20 --warmup 2'),
                       ('c3b', 'BASELINE configs[2]: 40 x 300 full batch, bf16', 'python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2')):
    open(os.path.join(P, f'r06_kernel_trace_{tag}.md'), 'w').write(
        f'# Round 6 (commit {head}) -- rocprofv3 --kernel-trace --stats, {what}\n\n'
        f'Command (tools/profile_r06.sh, on the MI355X box): `rocprofv3 --kernel-trace --stats -d ... -- {cmd} {B}`; per-step figures over '
        f'the graph replays of the timed region (tools/prof_summary.py), then one step\'s launches in time order per hardware queue '
        f'(tools/step_timeline.py; gap = start minus the previous end on the same queue, negative = overlapped).\n\n'
        + rd(f'summary_{tag}.md') + '\n## one step in time order\n```\n' + rd(f'timeline_{tag}.txt') + '```\n')


# ---- HBM-side traffic
def table(md):
    rows = []
    for l in md.splitlines():
        m = re.match(r'\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(5))))
    return rows


def step_bytes(rows, mesh_per_step):
    mesh = [r for r in rows if r[0].startswith('mesh_v2v_fused_kernel')]
    steps = sum(r[1] for r in mesh) / mesh_per_step
    tot = sum(n * (2 * f + w) for _, n, f, w in rows) / steps * 1048576
    return steps, tot, mesh


out = {}
text = ''
for key, f, mps, what in (('8x300x1xf32', 'pmc_traffic.md', 1, 'headline 8 x 300 fp32 (mesh_blend f32_split)'),
                          ('40x300x1xbf16', 'pmc_traffic_c3b.md', 2, 'C3 40 x 300 bf16')):
    md = rd(f)
    rows = table(md)
    steps, tot, mesh = step_bytes(rows, mps)
    mb = sum(n * (2 * fe + w) for _, n, fe, w in mesh) / sum(n for _, n, _, _ in mesh) * 1048576
    adj = [r for r in rows if 'gemm_adj' in r[0]]
    kern = {'mesh_v2v_fused': int(mb)}
    if key.endswith('f32'):      # (round 6: the blend-shape adjoint is the xp kernel's launch with the largest fetch: M x 207 x 20 670)
        xpk = sorted([r for r in rows if 'gemm_xp_kernel' in r[0]], key=lambda r: -r[2])
        if xpk:
            kern['gemm_xp_all_launches_mean'] = int((2 * xpk[0][2] + xpk[0][3]) * 1048576)
    out[key] = {'source': f'profiles/r06_pmc_traffic.md (tools/profile_r06.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, '
                          f'eager launches; FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE; all kernels of a step)',
                'commit': head, 'step_bytes': int(tot), 'kernels': kern, 'kernel_variants': {'mesh_v2v_fused': mesh[0][0]},
                'step_variant': mesh[0][0] + ('|mlp=f32_split' if key.endswith('f32') else '|mlp=bf16'),
                'body_model': 'synthetic, random vertex permutation (default)'}
    text += (f'## {what}\n\n{int(steps)} steps profiled; sum over the kernels of one step (launches / steps x (x2 fetch + write)): '
             f'**{tot / 1048576:.0f} MiB per step**; the mesh kernel `{mesh[0][0]}`: {mb / 1048576:.0f} MiB per launch.\n\n' + md + '\n')
open(os.path.join(P, 'r06_pmc_traffic.md'), 'w').write(
    f'# Round 6 (commit {head}) -- HBM-side traffic per kernel, separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)\n\n'
    f'Commands (tools/profile_r06.sh): `export NEMO_GRAPHS=0; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py --steps 4 '
    f'--warmup 1 {B}` and the same with `--pmc WRITE_SIZE`; for C3 with `--instances 40 --dtype bf16 --steps 3`.  Eager launches so that '
    f'every kernel is its own dispatch.  FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 '
    f'column is the corrected read volume for wide coalesced streams.  Infinity-Cache hits are counted (memory-side requests of the L2s).\n\n'
    + text)


# ---- MFMA pipe
def mfma(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
        by.setdefault((name, int(r['Grid_Size']) // 256), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for (name, blocks), v in by.items():
        m = {n: sum(x) / len(x) for n, x in v.items()}
        if not m.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            continue
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0
        rows.append((cyc * len(v['GRBM_GUI_ACTIVE']), name, blocks, len(v['GRBM_GUI_ACTIVE']), cyc, m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0),
                     m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512 / 1e9, m.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512 / 1e9))
    rows.sort(reverse=True)
    return rows


mt = ''
for key, d, what in (('8x300x1xf32', 'pmc_mfma', 'headline 8 x 300 fp32 (mesh_blend f32_split)'), ('40x300x1xbf16', 'pmc_mfma_c3b', 'C3 40 x 300 bf16')):
    rows = mfma(os.path.join(G, d, 'm_counter_collection.csv'))
    mt += (f'## {what}\n\n| kernel | blocks | launches | GPU cycles | MFMA pipe busy | executed GFLOP (fp32 MFMA) | executed GFLOP (bf16 MFMA) |\n'
           '|---|---:|---:|---:|---:|---:|---:|\n')
    for _, name, blocks, n, cyc, busy, f32, b16 in rows[:16]:
        mt += f'| `{name}` | {blocks} | {n} | {cyc:,.0f} | {100 * busy:.1f} % | {f32:.2f} | {b16:.2f} |\n'
    mt += '\n'
    mesh = [r for r in rows if r[1].startswith('mesh_v2v_fused_kernel')]
    if mesh:
        w = sum(r[3] * r[4] for r in mesh)
        out[key]['mfma_busy'] = {'mesh_v2v_fused': round(sum(r[5] * r[3] * r[4] for r in mesh) / w, 4)}
        out[key]['mfma_busy_source'] = 'profiles/r06_pmc_mfma.md (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs), cycle-weighted over the launches)'
open(os.path.join(P, 'r06_pmc_mfma.md'), 'w').write(
    f'# Round 6 (commit {head}) -- matrix-pipe counters per kernel, rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE '
    f'SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16\n\nCommands: tools/profile_r06.sh section 3 (eager launches).  MFMA pipe busy = '
    f'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); executed GFLOP = MOPS x 512 (the work the matrix pipes did, '
    f'padding and piece products included -- not the algorithmic FLOPs bench.py prices).\n\n' + mt)

# ---- mesh kernel LDS / L1 counters on the two body models
def lds(path):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
        by[name][r['Counter_Name']].append(float(r['Counter_Value']))
    return {n: {c: sum(x) / len(x) for c, x in v.items()} for n, v in by.items()}


lt = ''
for bm in ('default', 'locality'):
    try:
        rows = lds(os.path.join(G, f'pmc_lds_{bm}', 'l_counter_collection.csv'))
    except Exception as ex:
        lt += f'* {bm}: (no counters: {ex!r})\n'
        continue
    for name, m in rows.items():
        if name.startswith('mesh_v2v_fused_kernel'):
            act = m.get('SQ_LDS_IDX_ACTIVE', 0.0)
            lt += (f"| {bm} | `{name}` | {m.get('SQ_INSTS_LDS', 0):,.0f} | {act:,.0f} | {m.get('SQ_LDS_BANK_CONFLICT', 0):,.0f} | "
                   f"{100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(act, 1):.1f} % | {m.get('TCP_TCC_READ_REQ_sum', 0):,.0f} |\n")
try:
    loc = rd('summary_loc.md')
except Exception:
    loc = ''
open(os.path.join(P, 'r06_pmc_mesh_body_models.md'), 'w').write(
    f'# Round 6 (commit {head}) -- the mesh kernel on the two synthetic body models (VERDICT r05 item 8)\n\n'
    f'default = `synthetic.make_smpl_assets(6890, skin_nnz=4)`: vertices a random permutation, random joints per vertex, dense random J_regressor; '
    f'locality = `make_smpl_assets(..., locality=True)`: vertices ordered by body part, 1 - 2 dominant skinning weights among neighbouring joints, '
    f'sparse local joint regressors, pose blend shapes large only for the joints next to a vertex.  Commands: tools/profile_r06.sh section 1b '
    f'(`NEMO_BENCH_LOCALITY=1`), eager launches, `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCP_TCC_READ_REQ_sum`; '
    f'per-launch means over the run.\n\n| body model | kernel | LDS instructions | LDS active cycles | bank-conflict cycles | conflict / active | L1 -> L2 read requests |\n'
    f'|---|---|---:|---:|---:|---:|---:|\n' + lt + '\n## kernel trace on the locality model (graph replays)\n\n' + loc)

# the C4 entry of the previous round stays (the leg was not re-profiled)
old = json.load(open(os.path.join(P, 'traffic.json')))
for k, v in old.items():
    if k not in out:
        out[k] = v
json.dump(out, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)

# ---- bench lines
bl = f'# Round 6 (commit {head}) -- bench.py lines of tools/profile_r06.sh (one MI355X, un-profiled runs, {B.replace(" --repeat 1 --minibatch-steps 0", "")})\n\n'
for f, what in (('bench_c2_bf16.json', '8 x 300, gemm_dtype bf16'), ('bench_c3_f32.json', '40 x 300, fp32'), ('bench_c3_bf16.json', '40 x 300, bf16 (C3)'),
                ('bench_shard_v1.json', '1 x 300 (one rank\'s share of an 8-way split)'), ('bench_shard_v2.json', '2 x 300'), ('bench_shard_v4.json', '4 x 300'),
                ('bench_group1_v1.json', '1 x 300 through the sharded code path in a process group of one'),
                ('bench_group1_v8.json', '8 x 300 through the sharded code path in a process group of one')):
    try:
        d = json.loads(rd(f).strip().splitlines()[-1])
    except Exception as ex:
        bl += f'* {what}: (no line: {ex!r})\n'
        continue
    r = d.get('roofline') or {}
    bl += (f"* **{what}**: {d['value']} it/s, {d['ms_per_step']} ms/step, minibatch512 {((d.get('minibatch512') or {}).get('ms_per_step'))} ms; "
           f"`{r.get('kernel')}` {r.get('mean_launch_ms')} ms/launch, {r.get('achieved')} of {r.get('peak')} TFLOP/s (frac {r.get('frac')}), "
           f"mesh kernel {d['config'].get('mesh_kernel')}\n")
open(os.path.join(P, 'r06_bench_lines.md'), 'w').write(bl)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'source'} for k, v in out.items()}, indent=1)[:2500])
