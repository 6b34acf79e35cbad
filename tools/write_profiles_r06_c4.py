#!/usr/bin/env python3
"""gpurun_out/r06p/pmc_traffic_c4.md (tools/profile_r06_traffic_c4.sh) -> profiles/r06_pmc_traffic_c4.md + the C4 entry of profiles/traffic.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, 'gpurun_out', 'r06p'), os.path.join(ROOT, 'profiles')
head = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True, cwd=ROOT).stdout.split()[0]
md = open(os.path.join(G, 'pmc_traffic_c4.md')).read()
lk = {}
for l in md.splitlines():
    if l.startswith('| `'):
        c = [x.strip() for x in l.strip().strip('|').split('|')]
        lk[c[0].strip('`')] = (int(c[1]), float(c[3]), float(c[4]))
nst = [v[0] for k, v in lk.items() if k.startswith('adam_kernel')][0]
tot = sum(n / float(nst) * (f2 + w) for n, f2, w in lk.values())
mesh_name, mesh = [(k, v) for k, v in lk.items() if k.startswith('mesh_v2v_fused_kernel')][0]
t = json.load(open(os.path.join(P, 'traffic.json')))
t['256x1024x1xf32'] = {'source': 'profiles/r06_pmc_traffic_c4.md (tools/profile_r06_traffic_c4.sh: separate FETCH_SIZE / WRITE_SIZE passes of the leg, eager '
                                 'launches; FETCH_SIZE x2 + WRITE_SIZE)', 'commit': head, 'step_bytes': int(tot * 2 ** 20),
                       'kernels': {'mesh_v2v_fused': int((mesh[1] + mesh[2]) * 2 ** 20)}, 'kernel_variants': {'mesh_v2v_fused': mesh_name},
                       'step_variant': mesh_name + '|mlp=f32_split', 'body_model': 'synthetic, random vertex permutation (default)'}
json.dump(t, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
open(os.path.join(P, 'r06_pmc_traffic_c4.md'), 'w').write(
    f'# Round 6 (commit {head}) -- HBM-side traffic per kernel of the C4 leg (256 x 1024 full batch, fp32 defaults: mesh_blend / mlp_gemm f32_split)\n\n'
    f'`bench.py --instances 256 --frames 1024 --steps 2 --warmup 1` under `NEMO_GRAPHS=0`, separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; '
    f'{nst} steps.  Sum over the kernels of one step: **{tot / 1024:.1f} GiB = {tot * 2 ** 20 / 1e9:.1f} GB** (x2-corrected fetch + write; memory-side '
    f'requests of the L2s: Infinity-Cache hits are counted).  Round 5: 165.9 GB.\n\n' + md)
print(t['256x1024x1xf32'])
