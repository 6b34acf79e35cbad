#!/usr/bin/env python3
"""Turns what `gpurun -- 'bash tools/profile_r02.sh'` left under gpurun_out/r02p/ into the committed round-2 summaries
under profiles/ and refreshes profiles/traffic.json (the counter-measured bytes bench.py quotes)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out', 'r02p')
P = os.path.join(ROOT, 'profiles')
head = subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True, cwd=ROOT).stdout.split()[0]
rd = lambda f: open(os.path.join(G, f)).read()
line = lambda f: json.loads([l for l in rd(f).splitlines() if l.startswith('{')][-1])
B = '--no-cpu-baseline --no-torch-gpu-baseline'


def bench_row(name, f, cmd):
    d = line(f)
    r = d['roofline']
    return (f"| {name} | `{cmd}` | {d['value']} | {d['ms_per_step']} | {d['dtype']} | {r['kernel']} {r['mean_launch_ms']} ms, "
            f"{r['achieved']} TFLOP/s = {r['frac']} of {r['peak']} | {r['step']['achieved']} TFLOP/s = {r['step']['frac']} |")


# ---- kernel traces
for tag, title, cmd, steps in (
        ('c2', 'headline C2 (8 x 300 full batch, fp32)', f'python3 bench.py --steps 20 --warmup 2 {B}', 35),
        ('v1', 'one-instance shard (1 x 300: one rank of eight)', f'python3 bench.py --instances 1 --steps 20 --warmup 2 {B}', 35),
        ('c3', 'C3 (40 x 300) fp32', f'python3 bench.py --instances 40 --steps 10 --warmup 2 {B}', 25),
        ('c3b', 'C3 (40 x 300) bf16 dense contractions', f'python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 {B}', 25)):
    d = line(f'trace_{tag}.log')
    r = d['roofline']
    tl = rd(f'timeline_{tag}.txt')
    open(os.path.join(P, f'r02_kernel_trace_{tag}.md'), 'w').write(
        f"# Round 2 (commit {head}) -- rocprofv3 --kernel-trace --stats, {title}\n\n"
        f"Command (on the MI355X box): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats "
        f"-d ... -- {cmd}` ({steps} steps traced: set-up + warm-up + timed graph replays + instrumented eager steps)\n\n"
        f"bench line of the same (profiled) run: {d['value']} it/s, {d['ms_per_step']} ms/step; roofline kernel {r['kernel']} "
        f"{r['mean_launch_ms']} ms/launch (HIP events in bench.py) -> {r['achieved']} TFLOP/s = {r['frac']} of the {r['peak']} "
        f"TFLOP/s {d['dtype']} MFMA peak; whole step {r['step']['achieved']} TFLOP/s of algorithmic work\n\n"
        + rd(f'summary_{tag}.md') + "\n## one graph-replayed step (tools/step_timeline.py: start, duration, gap to the previous kernel's end; "
        "q = hardware queue)\n```\n" + tl + "```\n")

# ---- traffic
pmc = rd('pmc_traffic.md')
rows = [l for l in pmc.splitlines() if l.startswith('| `')]
tot, kern = 0.0, {}
for l in rows:
    c = [x.strip() for x in l.strip().strip('|').split('|')]
    n, f2, w = int(c[1]), float(c[3]), float(c[4])
    tot += n / 12.0 * (f2 + w)
    kern[c[0].strip('`')] = (n, f2, w)
mesh = kern['mesh_v2v_fused_kernel<false>']
adj = [v for k, v in kern.items() if 'false, true, 3, true, false' in k][0]
traffic = {'8x300x1xf32': {
    'source': 'profiles/r02_pmc_traffic.md (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, eager launches; '
              'FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE; the 24 largest kernels of a step)',
    'step_bytes': int(tot * 2 ** 20),
    'kernels': {'mesh_v2v_fused': int((mesh[1] + mesh[2]) * 2 ** 20), 'gemm_pose_blend_bwd': int((adj[1] + adj[2]) * 2 ** 20)}}}
json.dump(traffic, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
open(os.path.join(P, 'r02_pmc_traffic.md'), 'w').write(
    f"# Round 2 (commit {head}) -- HBM-side traffic per kernel, separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)\n\n"
    f"Commands: `export NEMO_GRAPHS=0; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py --steps 4 --warmup 1 {B}` "
    "and the same with `--pmc WRITE_SIZE` (eager launches so that every kernel is its own dispatch; 12 steps in total).\n"
    "FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 column is the corrected read "
    "volume for wide coalesced streams.  Infinity-Cache hits are counted (memory-side requests of the L2s).\n\n"
    f"Sum over the kernels of one step (launches / 12 x (x2 fetch + write)): **{tot:.0f} MiB per step** -> at the "
    f"{line('bench_c2_full.json')['ms_per_step']} ms step of the un-profiled run {tot * 2 ** 20 / line('bench_c2_full.json')['ms_per_step'] / 1e6:.0f} GB/s "
    "= the `roofline.hbm` entry of the bench line (profiles/traffic.json).\n"
    "The blend-shape adjoint GEMM (`gemm_glds_kernel<..., false, true, ...>`, K = 20670) still re-reads the 189 MiB transposed dVP "
    "once per 64-column tile of its 207 outputs (4x): the two alternatives built this round -- a 64 x 208 tile (one pass over "
    "dVP^T) and folding the adjoint into the mesh kernel (no dVP at all) -- both measured no faster, see "
    "profiles/r02_experiments.md.\n\n" + pmc)

# ---- MFMA
mf = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_mfma_summary.py'), G], capture_output=True, text=True).stdout
open(os.path.join(P, 'r02_pmc_mfma.md'), 'w').write(
    f"# Round 2 (commit {head}) -- MFMA-pipe and LDS counters per kernel, separate rocprofv3 --pmc passes\n\n"
    f"Commands (eager launches): `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 "
    f"--output-format csv -- python3 bench.py --steps 4 --warmup 1 {B}` and `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES`.\n"
    "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); executed GFLOP = MOPS_F32 x 512.\n"
    "`glds::gemm_glds_kernel<64, 64, 32, 32, 32, AKC, BKC, 3, true, false>`: AKC / BKC = operand is k-contiguous in memory "
    "(true, true = NT: nn.Linear forward; true, false = NN: activation gradients; false, false = TN: parameter gradients; "
    "false, true = TT: blend-shape adjoint).\n\n" + mf)

# ---- bench lines
full = line('bench_c2_full.json')
open(os.path.join(P, 'r02_bench_lines.md'), 'w').write(
    f"# Round 2 (commit {head}) -- bench.py lines of every BASELINE configuration on one MI355X (un-profiled runs of one gpurun call)\n\n"
    "| configuration | command | it/s | ms/step | dtype | roofline kernel | whole step (algorithmic FLOPs / step time vs fp32 MFMA peak) |\n"
    "|---|---|---:|---:|---|---|---|\n"
    + bench_row('C2 headline: 8 x 300 full batch', 'bench_c2_full.json', 'python3 bench.py') + '\n'
    + bench_row('C2 sizes, bf16 dense contractions', 'bench_c2_bf16.json', f'python3 bench.py --dtype bf16 --steps 30 --warmup 5') + '\n'
    + bench_row('C3: 40 x 300, fp32', 'bench_c3_f32.json', 'python3 bench.py --instances 40 --steps 20 --warmup 3') + '\n'
    + bench_row('C3: 40 x 300, bf16 (BASELINE configs[2])', 'bench_c3_bf16.json', 'python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3') + '\n'
    + bench_row('C4: 256 x 1024 on ONE GPU (32 mesh chunks of 8192)', 'bench_c4.json', 'python3 bench.py --instances 256 --frames 1024 --steps 5 --warmup 2') + '\n'
    + bench_row('shard of 8 GPUs: 1 x 300', 'bench_shard_v1.json', 'python3 bench.py --instances 1 --steps 100') + '\n'
    + bench_row('shard of 4 GPUs: 2 x 300', 'bench_shard_v2.json', 'python3 bench.py --instances 2 --steps 100') + '\n'
    + bench_row('shard of 2 GPUs: 4 x 300', 'bench_shard_v4.json', 'python3 bench.py --instances 4 --steps 100') + '\n'
    + "\n(all but the first with `--no-cpu-baseline --no-torch-gpu-baseline`; with `--dtype bf16` the roofline kernel is priced "
      "against the 2.5 PFLOP/s dense bf16 peak although only its pose blend runs there -- skinning, L1 and the adjoints stay on the "
      "fp32 pipe, which bounds it.)\n\n## the full default line (what the driver records)\n```\n" + json.dumps(full) + "\n```\n"
      f"cpu_baseline: {json.dumps(full['cpu_baseline'])}\n")

# ---- GEMM harness
open(os.path.join(P, 'r02_gemm_glds.md'), 'w').write(
    f"# Round 2 (commit {head}) -- the LDS-DMA GEMM core (csrc/gemm_glds.h) in isolation: tools/gemm_glds_dev\n\n"
    "`calib`: 256 r tiles of 64x64 (r blocks per CU), `it` K tiles each, no split, constant operands -- microseconds per launch; the "
    "slope in `it` is the K-tile time of r co-resident blocks (`64x64 spread` = pieces of the next tile issued between the MFMAs, "
    "the product kernel; `64x64` = all pieces at the top of the iteration).\n```\n" + rd('gemm_calib.txt') + "```\n"
    "`time 2400`: the step's shapes at N = 2400, random operands, every tile configuration x K split (`sN`; `t` = whole tiles + "
    "split tail).  fp32 MFMA peak 157.3 TFLOP/s.\n```\n" + rd('gemm_time_2400.txt') + "```\n"
    "`time 300`: one rank of eight\n```\n" + rd('gemm_time_300.txt') + "```\n")
print('wrote profiles for', head, '; step traffic MiB', round(tot))
