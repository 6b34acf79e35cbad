#!/usr/bin/env python3
"""Turns what `gpurun -- 'bash tools/profile_r04.sh'` left under gpurun_out/r04p/ into the committed round-4 summaries under
profiles/ and refreshes profiles/traffic.json (the counter-measured bytes bench.py quotes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out', 'r04p')
P = os.path.join(ROOT, 'profiles')
head = subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True, cwd=ROOT).stdout.split()[0]
rd = lambda f: open(os.path.join(G, f)).read()
line = lambda f: json.loads([l for l in rd(f).splitlines() if l.startswith('{')][-1])
B = '--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs'
X = '--repeat 1 --minibatch-steps 0'


def bench_row(name, f, cmd, d=None):
    d = d or line(f)
    r = d['roofline']
    mb = d.get('minibatch512')
    rep = d.get('repeat_ms_per_step')
    rng = f" ({min(rep)} - {max(rep)})" if rep else ''
    return (f"| {name} | `{cmd}` | {d['value']} | {d['ms_per_step']}{rng} | {d['dtype']} | "
            f"{r['kernel']} {r['mean_launch_ms']} ms, {r['achieved']} TFLOP/s = {r['frac']} of {r['peak']} | "
            f"{r['step']['achieved']} TFLOP/s = {r['step']['frac']} of {r['step']['peak']} | "
            f"{(str(mb['value']) + ' it/s, ' + str(mb['ms_per_step']) + ' ms, replay ' + str(mb['graph_replay_fraction'])) if mb else '-'} |")


# ---- kernel traces
for tag, title, cmd in (
        ('c2', 'headline C2 (8 x 300 full batch, fp32)', f'python3 bench.py --steps 20 --warmup 2 {X} {B}'),
        ('v1', 'one-instance shard (1 x 300: one rank of eight)', f'python3 bench.py --instances 1 --steps 20 --warmup 2 {X} {B}'),
        ('c3b', 'C3 (40 x 300) bf16, operands bf16 in memory', f'python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 {X} {B}')):
    if not os.path.exists(os.path.join(G, f'trace_{tag}.log')):
        continue
    d = line(f'trace_{tag}.log')
    r = d['roofline']
    open(os.path.join(P, f'r04_kernel_trace_{tag}.md'), 'w').write(
        f"# Round 4 (commit {head}) -- rocprofv3 --kernel-trace --stats, {title}\n\n"
        f"Command (on the MI355X box): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats "
        f"-d ... -- {cmd}` (set-up + warm-up + timed graph replays + instrumented eager steps are all in the trace)\n\n"
        f"bench line of the same (profiled) run: {d['value']} it/s, {d['ms_per_step']} ms/step; roofline kernel {r['kernel']} "
        f"{r['mean_launch_ms']} ms/launch (HIP events in bench.py, un-captured launches) -> {r['achieved']} TFLOP/s = {r['frac']} of the "
        f"{r['peak']} TFLOP/s {d['dtype']} MFMA peak; whole step {r['step']['achieved']} TFLOP/s of algorithmic work.  The rocprofv3 average of "
        f"the same kernel is in the table below (`avg us/launch`).\n\n"
        + rd(f'summary_{tag}.md') + "\n## one graph-replayed step (tools/step_timeline.py: start, duration, gap to the previous kernel's end; "
        "q = hardware queue).  Cross-queue latencies are over-stated under the profiler: the un-profiled per-kernel contributions are in "
        "r04_bench_lines.md (ablation)\n```\n" + rd(f'timeline_{tag}.txt') + "```\n")
if os.path.exists(os.path.join(G, 'trace_phases.log')):
    open(os.path.join(P, 'r04_kernel_trace_phases.md'), 'w').write(
        f"# Round 4 (commit {head}) -- rocprofv3 --kernel-trace --stats of the fit phases (tools/bench_phases.py)\n\n"
        "20 + 300 warm-up iterations (batch 512), 20 + 1000 camera-fit iterations (8 views), 20 + 500 minibatch-512 steps of the published "
        "configuration, 8 x 300 synthetic sequences.  Warm-up and camera fit run as captured iterations (DESIGN.md 5a): the kernels below "
        "with 1320 / 1020 calls are theirs (`seq_gather_kernel`, `seq_log_kernel`, `nan_count_kernel`, `step_begin_kernel`; the camera fit has no "
        "GEMM and no FK in its loop).\n\nWall clock of the same (profiled) run:\n```\n"
        + '\n'.join(l for l in rd('trace_phases.log').splitlines() if l.startswith(('warmup', 'opt_cam', 'minibatch')) or 'ms/step' in l)
        + "\n```\n(un-profiled: 0.259 / 0.033 / 0.645 ms -- the profiler's per-dispatch cost is large against 5 - 20 us kernels.)\n\n"
        "Per-kernel totals over the WHOLE run (the `calls/step` and `us/step` columns are totals divided by 35, the script's divisor: read "
        "them as relative weights; `avg us/launch` is exact):\n\n" + rd('summary_phases.md'))

# ---- traffic
pmc = rd('pmc_traffic.md')
rows = [l for l in pmc.splitlines() if l.startswith('| `')]
tot, kern = 0.0, {}
for l in rows:
    c = [x.strip() for x in l.strip().strip('|').split('|')]
    kern[c[0].strip('`')] = (int(c[1]), float(c[3]), float(c[4]))
mesh = [v for k, v in kern.items() if k.startswith('mesh_v2v_fused_kernel<0')][0]
nsteps = mesh[0]                                        # one mesh launch per step
for k, (n, f2, w) in kern.items():
    tot += n / float(nsteps) * (f2 + w)
adj = [v for k, v in kern.items() if 'gemm_adj_kernel' in k][0]
full = line('bench_c2_full.json')
traffic = {'8x300x1xf32': {
    'source': 'profiles/r04_pmc_traffic.md (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, eager launches; '
              'FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE; the largest kernels of a step)',
    'step_bytes': int(tot * 2 ** 20),
    'kernels': {'mesh_v2v_fused': int((mesh[1] + mesh[2]) * 2 ** 20), 'gemm_pose_blend_bwd': int((adj[1] + adj[2]) * 2 ** 20)}}}
# the C3 bf16 and C4 legs (tools/profile_r04_traffic_legs.sh), when collected
legs_txt = ''
for tag, key, title, cmd in (
        ('c3b', '40x300x1xbf16', 'C3 (40 x 300) bf16', '--instances 40 --dtype bf16 --steps 3 --warmup 1'),
        ('c4', '256x1024x1xf32', 'C4 (256 x 1024) fp32', '--instances 256 --frames 1024 --steps 2 --warmup 1')):
    f = os.path.join(G, f'pmc_traffic_{tag}.md')
    if not os.path.exists(f):
        continue
    lk = {}
    for l in open(f).read().splitlines():
        if l.startswith('| `'):
            c = [x.strip() for x in l.strip().strip('|').split('|')]
            lk[c[0].strip('`')] = (int(c[1]), float(c[3]), float(c[4]))
    nst = [v[0] for k, v in lk.items() if k.startswith('adam_kernel')][0]          # one Adam launch per step
    ltot = sum(n / float(nst) * (f2 + wv) for n, f2, wv in lk.values())
    lmesh = [v for k, v in lk.items() if k.startswith('mesh_v2v_fused_kernel')][0]
    traffic[key] = {'source': 'profiles/r04_pmc_traffic_legs.md (tools/profile_r04_traffic_legs.sh: separate FETCH_SIZE / WRITE_SIZE '
                              'passes of the leg, eager launches; FETCH_SIZE x2 + WRITE_SIZE)',
                    'step_bytes': int(ltot * 2 ** 20), 'kernels': {'mesh_v2v_fused': int((lmesh[1] + lmesh[2]) * 2 ** 20)}}
    legs_txt += (f"## {title}: `bench.py {cmd} --repeat 1 --minibatch-steps 0 {B}` under `NEMO_GRAPHS=0`, {nst} steps\n\n"
                 f"Sum over the kernels of one step: **{ltot:.0f} MiB** (x2-corrected fetch + write; memory-side requests of the L2s: "
                 "Infinity-Cache hits are counted, so re-reads of an operand that lives in the 256 MB cache show up here without "
                 f"reaching HBM).\n\n" + open(f).read() + '\n')
if legs_txt:
    open(os.path.join(P, 'r04_pmc_traffic_legs.md'), 'w').write(
        f"# Round 4 (commit {head}) -- HBM-side traffic per kernel of the C3 bf16 and C4 legs (separate --pmc passes)\n\n" + legs_txt)
json.dump(traffic, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
open(os.path.join(P, 'r04_pmc_traffic.md'), 'w').write(
    f"# Round 4 (commit {head}) -- HBM-side traffic per kernel, separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)\n\n"
    f"Commands: `export NEMO_GRAPHS=0; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py --steps 4 --warmup 1 {X} {B}` "
    f"and the same with `--pmc WRITE_SIZE` (eager launches so that every kernel is its own dispatch; {nsteps} steps in total).\n"
    "FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 column is the corrected read "
    "volume for wide coalesced streams.  Infinity-Cache hits are counted (memory-side requests of the L2s).\n\n"
    f"Sum over the kernels of one step (launches / {nsteps} x (x2 fetch + write)): **{tot:.0f} MiB per step** -> at the "
    f"{full['ms_per_step']} ms step of the un-profiled run {tot * 2 ** 20 / full['ms_per_step'] / 1e6:.0f} GB/s "
    "= the `roofline.hbm` entry of the bench line (profiles/traffic.json).  Round 3: 2746 MiB; the difference is the blend-shape adjoint "
    f"(`glds::gemm_adj_kernel`: {adj[1]:.0f} MiB x2-corrected fetch + {adj[2]:.0f} MiB written per launch against 939 + 14 for the 64 x 64 plan: "
    "`dVP^T` is streamed once).\n\n" + pmc)

# ---- MFMA
mf = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_mfma_summary.py'), G], capture_output=True, text=True).stdout
open(os.path.join(P, 'r04_pmc_mfma.md'), 'w').write(
    f"# Round 4 (commit {head}) -- MFMA-pipe counters per kernel, separate rocprofv3 --pmc pass\n\n"
    f"Command (eager launches): `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 "
    f"--output-format csv -- python3 bench.py --steps 4 --warmup 1 {X} {B}`.\n"
    "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); executed GFLOP = MOPS_F32 x 512.\n"
    "`glds::gemm_glds_kernel<64, 64, 32, 32, 32, AKC, BKC, 3, true, 0>`: AKC / BKC = operand is k-contiguous in memory "
    "(true, true = NT: nn.Linear forward; true, false = NN: activation gradients; false, false = TN: parameter gradients); "
    "`glds::gemm_adj_kernel` = the blend-shape adjoint on the mixed-shape 64 x 208 tile (round 3: the TT instance of the 64 x 64 kernel, "
    "71 % busy with 25.7 GFLOP executed for 20.5 algorithmic).\n\n" + mf)

# ---- bench lines
hdr = ("| configuration | command | it/s | ms/step (min - max of the timed regions) | dtype | roofline kernel | whole step (algorithmic FLOPs / "
       "step time vs the peak of its mix of pipes) | random minibatches of 512 |\n|---|---|---:|---:|---|---|---|---|\n")
c3b, c4 = dict(full['c3_bf16']), dict(full['c4'])
txt = (f"# Round 4 (commit {head}) -- bench.py lines on one MI355X (un-profiled runs of one gpurun call)\n\n"
       "`value` = the MEDIAN of `--repeat` (default 5) timed regions of `--steps` steps each.  Since round 4 the DEFAULT line carries "
       "every BASELINE configuration: `c3_bf16`, `c4`, `published_fit`, `scaling_model` are keys of the first row's line.\n\n" + hdr
       + bench_row('C2 headline: 8 x 300 full batch', 'bench_c2_full.json', 'python3 bench.py') + '\n'
       + bench_row('C3: 40 x 300, bf16 (BASELINE configs[2]) -- key `c3_bf16` of the default line', None, 'python3 bench.py', c3b) + '\n'
       + bench_row('C4: 256 x 1024 on ONE GPU (32 mesh chunks of 8192) -- key `c4` of the default line', None, 'python3 bench.py', c4) + '\n'
       + bench_row('C2 sizes, bf16 dense contractions', 'bench_c2_bf16.json', f'python3 bench.py --dtype bf16 --steps 30 --warmup 5 {B}') + '\n'
       + bench_row('C3: 40 x 300, fp32', 'bench_c3_f32.json', f'python3 bench.py --instances 40 --steps 20 --warmup 3 {B}') + '\n'
       + bench_row('shard of 8 GPUs: 1 x 300', 'bench_shard_v1.json', f'python3 bench.py --instances 1 --steps 100 {B}') + '\n'
       + bench_row('shard of 4 GPUs: 2 x 300', 'bench_shard_v2.json', f'python3 bench.py --instances 2 --steps 100 {B}') + '\n'
       + bench_row('shard of 2 GPUs: 4 x 300', 'bench_shard_v4.json', f'python3 bench.py --instances 4 --steps 100 {B}') + '\n'
       + "\n## published schedule end to end (`published_fit` of the default line; nemomocap-example.sh:10,17,30-33 through fit.run_fit)\n```\n"
       + json.dumps(full['published_fit'], indent=1) + "\n```\n"
       + "\n## scaling model (`scaling_model` of the default line: one rank's share measured on this GPU)\n```\n"
       + json.dumps(full['scaling_model'], indent=1) + "\n```\n"
       + "\n## the sharded code path in a process group of ONE rank (`NEMO_BENCH_SHARD_OF_ONE=1`: ShardedNemo, RCCL communicator of world size 1; "
         "collectives inside the step's HIP graph)\n\n"
         "| instances | it/s | ms/step | modes timed by `--shard-mode auto` (ms/step) | kept | same loss after 14 steps from the same state | compute_ms (collectives skipped) | collective_ms (9 MB all-reduce alone, world of one) |\n"
         "|---:|---:|---:|---|---|---|---:|---:|\n")
for v in (1, 8):
    fn = f'bench_group1_v{v}.json'
    if not os.path.exists(os.path.join(G, fn)) or not rd(fn).strip():
        continue
    d = line(fn)
    pr = d['per_rank'][0]
    txt += (f"| {v} | {d['value']} | {d['ms_per_step']} | {json.dumps(d['shard_modes_ms'])} | {d['shard_mode']} | "
            f"{all(c['agrees_with_single'] for c in d['shard_mode_check'].values())} | {pr['compute_ms']} | {pr['collective_ms']} |\n")
if os.path.exists(os.path.join(G, 'ablate_v8.txt')):
    txt += ("\n## un-profiled contribution of single kernels to the 8 x 300 step (`bash tools/ablate.sh 8 ...`: bench.py with ONE C entry point "
            "turned into a no-op by tools/ablate.py; delta = what the kernel contributes to the replayed graph's critical path)\n```\n"
            + '\n'.join(l for l in rd('ablate_v8.txt').splitlines() if l.startswith(('full step', 'nemo_'))) + "\n```\n")
for f, t in (('adj_2400.txt', 'M = 2400'), ('adj_1200.txt', 'M = 1200'), ('adj_8192.txt', 'M = 8192')):
    if os.path.exists(os.path.join(G, f)):
        txt += (f"\n## blend-shape adjoint harness, {t} (`tools/gemm_glds_dev adj`, us per launch; sN = N K slices)\n```\n"
                + '\n'.join(l for l in rd(f).splitlines() if not l.startswith('adj 64x208')) + "\n```\n")
txt += ("\n## the full default line (what the driver records)\n```\n" + json.dumps(full) + "\n```\n")
open(os.path.join(P, 'r04_bench_lines.md'), 'w').write(txt)
print('wrote profiles for', head, '; step traffic MiB', round(tot))
