#!/usr/bin/env python3
"""Turns what `gpurun -- 'bash tools/profile_r03.sh'` left under gpurun_out/r03p/ (+ the ablation / harness logs of the
round's other gpurun calls under gpurun_out/) into the committed round-3 summaries under profiles/ and refreshes
profiles/traffic.json (the counter-measured bytes bench.py quotes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out', 'r03p')
GO = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
head = subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True, cwd=ROOT).stdout.split()[0]
rd = lambda f: open(os.path.join(G, f)).read()
line = lambda f: json.loads([l for l in rd(f).splitlines() if l.startswith('{')][-1])
B = '--no-cpu-baseline --no-torch-gpu-baseline'
X = '--repeat 1 --minibatch-steps 0'


def bench_row(name, f, cmd):
    d = line(f)
    r = d['roofline']
    mb = d.get('minibatch512')
    return (f"| {name} | `{cmd}` | {d['value']} | {d['ms_per_step']} ({min(d['repeat_ms_per_step'])} - {max(d['repeat_ms_per_step'])}) | {d['dtype']} | "
            f"{r['kernel']} {r['mean_launch_ms']} ms, {r['achieved']} TFLOP/s = {r['frac']} of {r['peak']} | "
            f"{r['step']['achieved']} TFLOP/s = {r['step']['frac']} of {r['step']['peak']} | "
            f"{(str(mb['value']) + ' it/s, ' + str(mb['ms_per_step']) + ' ms, replay ' + str(mb['graph_replay_fraction'])) if mb else '-'} |")


# ---- kernel traces
for tag, title, cmd in (
        ('c2', 'headline C2 (8 x 300 full batch, fp32)', f'python3 bench.py --steps 20 --warmup 2 {X} {B}'),
        ('v1', 'one-instance shard (1 x 300: one rank of eight)', f'python3 bench.py --instances 1 --steps 20 --warmup 2 {X} {B}'),
        ('v2', 'two-instance shard (2 x 300: one rank of four)', f'python3 bench.py --instances 2 --steps 20 --warmup 2 {X} {B}'),
        ('c3b', 'C3 (40 x 300) bf16, operands bf16 in memory', f'python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 {X} {B}'),
        ('g1', 'the SHARDED step (1 x 300) in an RCCL process group of one rank, `split` mode: weighting of the loss scalars, the all-reduces '
               '(a world of one reduces in place: no kernel), the hand-over and the fused Adam inside the step\'s one graph',
         f'NEMO_BENCH_SHARD_OF_ONE=1 python3 bench.py --instances 1 --shard-mode split --steps 20 --warmup 2 {X} {B}')):
    if not os.path.exists(os.path.join(G, f'trace_{tag}.log')):
        continue
    d = line(f'trace_{tag}.log')
    r = d['roofline']
    open(os.path.join(P, f'r03_kernel_trace_{tag}.md'), 'w').write(
        f"# Round 3 (commit {head}) -- rocprofv3 --kernel-trace --stats, {title}\n\n"
        f"Command (on the MI355X box): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats "
        f"-d ... -- {cmd}` (set-up + warm-up + timed graph replays + instrumented eager steps are all in the trace)\n\n"
        f"bench line of the same (profiled) run: {d['value']} it/s, {d['ms_per_step']} ms/step; roofline kernel {r['kernel']} "
        f"{r['mean_launch_ms']} ms/launch (HIP events in bench.py, un-captured launches) -> {r['achieved']} TFLOP/s = {r['frac']} of the "
        f"{r['peak']} TFLOP/s {d['dtype']} MFMA peak; whole step {r['step']['achieved']} TFLOP/s of algorithmic work.  The rocprofv3 average of "
        f"the same kernel is in the table below (`avg us/launch`).\n\n"
        + rd(f'summary_{tag}.md') + "\n## one graph-replayed step (tools/step_timeline.py: start, duration, gap to the previous kernel's end; "
        "q = hardware queue).  Cross-queue latencies are over-stated under the profiler: the un-profiled per-kernel contributions are in "
        "r03_shard_budget.md\n```\n" + rd(f'timeline_{tag}.txt') + "```\n")

# ---- traffic
pmc = rd('pmc_traffic.md')
rows = [l for l in pmc.splitlines() if l.startswith('| `')]
tot, kern = 0.0, {}
nsteps = 4 + 1 + 3 + 3 + 1     # --steps 4 --warmup 1: 3 set-up + 1 warm-up + 4 timed + 3 instrumented ... (launch counts / steps below)
for l in rows:
    c = [x.strip() for x in l.strip().strip('|').split('|')]
    kern[c[0].strip('`')] = (int(c[1]), float(c[3]), float(c[4]))
mesh = [v for k, v in kern.items() if k in ('mesh_v2v_fused_kernel<false>', 'mesh_v2v_fused_kernel<0>')][0]
nsteps = mesh[0]                                        # one mesh launch per step
for k, (n, f2, w) in kern.items():
    tot += n / float(nsteps) * (f2 + w)
adj = [v for k, v in kern.items() if 'false, true, 3, true, false' in k or 'false, true, 3, true, 0>' in k][0]
full = line('bench_c2_full.json')
traffic = {'8x300x1xf32': {
    'source': 'profiles/r03_pmc_traffic.md (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, eager launches; '
              'FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE; the largest kernels of a step)',
    'step_bytes': int(tot * 2 ** 20),
    'kernels': {'mesh_v2v_fused': int((mesh[1] + mesh[2]) * 2 ** 20), 'gemm_pose_blend_bwd': int((adj[1] + adj[2]) * 2 ** 20)}}}
json.dump(traffic, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
open(os.path.join(P, 'r03_pmc_traffic.md'), 'w').write(
    f"# Round 3 (commit {head}) -- HBM-side traffic per kernel, separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)\n\n"
    f"Commands: `export NEMO_GRAPHS=0; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py --steps 4 --warmup 1 {X} {B}` "
    f"and the same with `--pmc WRITE_SIZE` (eager launches so that every kernel is its own dispatch; {nsteps} steps in total).\n"
    "FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 column is the corrected read "
    "volume for wide coalesced streams.  Infinity-Cache hits are counted (memory-side requests of the L2s).\n\n"
    f"Sum over the kernels of one step (launches / {nsteps} x (x2 fetch + write)): **{tot:.0f} MiB per step** -> at the "
    f"{full['ms_per_step']} ms step of the un-profiled run {tot * 2 ** 20 / full['ms_per_step'] / 1e6:.0f} GB/s "
    "= the `roofline.hbm` entry of the bench line (profiles/traffic.json).\n\n" + pmc)

# ---- MFMA
mf = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_mfma_summary.py'), G], capture_output=True, text=True).stdout
open(os.path.join(P, 'r03_pmc_mfma.md'), 'w').write(
    f"# Round 3 (commit {head}) -- MFMA-pipe counters per kernel, separate rocprofv3 --pmc pass\n\n"
    f"Command (eager launches): `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 "
    f"--output-format csv -- python3 bench.py --steps 4 --warmup 1 {X} {B}`.\n"
    "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); executed GFLOP = MOPS_F32 x 512.\n"
    "`glds::gemm_glds_kernel<64, 64, 32, 32, 32, AKC, BKC, 3, true, 0>`: AKC / BKC = operand is k-contiguous in memory "
    "(true, true = NT: nn.Linear forward; true, false = NN: activation gradients; false, false = TN: parameter gradients; "
    "false, true = TT: blend-shape adjoint).\n\n" + mf)

# ---- bench lines
hdr = ("| configuration | command | it/s | ms/step (min - max of the timed regions) | dtype | roofline kernel | whole step (algorithmic FLOPs / "
       "step time vs the peak of its mix of pipes) | random minibatches of 512 |\n|---|---|---:|---:|---|---|---|---|\n")
txt = (f"# Round 3 (commit {head}) -- bench.py lines of every BASELINE configuration on one MI355X (un-profiled runs of one gpurun call)\n\n"
       "`value` = the MEDIAN of `--repeat` (default 5) timed regions of `--steps` steps each.\n\n" + hdr
       + bench_row('C2 headline: 8 x 300 full batch', 'bench_c2_full.json', 'python3 bench.py') + '\n'
       + bench_row('C2 sizes, bf16 dense contractions', 'bench_c2_bf16.json', 'python3 bench.py --dtype bf16 --steps 30 --warmup 5') + '\n'
       + bench_row('C3: 40 x 300, fp32', 'bench_c3_f32.json', 'python3 bench.py --instances 40 --steps 20 --warmup 3') + '\n'
       + bench_row('C3: 40 x 300, bf16 (BASELINE configs[2])', 'bench_c3_bf16.json', 'python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3') + '\n'
       + (bench_row('C3 bf16 with NEMO_MESH_SPLIT=0 (mesh kernel\'s vertex->joint adjoint on the fp32 pipe, same box)', 'bench_c3_bf16_nosplit.json', 'NEMO_MESH_SPLIT=0 python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3') + '\n' if os.path.exists(os.path.join(G, 'bench_c3_bf16_nosplit.json')) else '')
       + (bench_row('C3 bf16 with NEMO_BF16_MEM=0 (round 2\'s on-the-fly rounding, same box)', 'bench_c3_bf16_onthefly.json', 'NEMO_BF16_MEM=0 python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3') + '\n' if os.path.exists(os.path.join(G, 'bench_c3_bf16_onthefly.json')) else '')
       + bench_row('C4: 256 x 1024 on ONE GPU (32 mesh chunks of 8192)', 'bench_c4.json',
                   'python3 bench.py --instances 256 --frames 1024 --steps 5 --warmup 2 --repeat 3 --minibatch-steps 20') + '\n'
       + bench_row('shard of 8 GPUs: 1 x 300', 'bench_shard_v1.json', 'python3 bench.py --instances 1 --steps 100') + '\n'
       + bench_row('shard of 4 GPUs: 2 x 300', 'bench_shard_v2.json', 'python3 bench.py --instances 2 --steps 100') + '\n'
       + bench_row('shard of 2 GPUs: 4 x 300', 'bench_shard_v4.json', 'python3 bench.py --instances 4 --steps 100') + '\n'
       + "\n(all but the first with `--no-cpu-baseline --no-torch-gpu-baseline`.  `--dtype bf16`: each part of a kernel is priced against the "
         "pipe it runs on -- the quoted peak is the harmonic mix of the bf16 (2500) and fp32 (157.3) MFMA peaks over the kernel's own GFLOP per "
         "pipe; the fused mesh kernel keeps its skinning / L1 on the fp32 pipe, which bounds it.)\n\n"
         "## the sharded code path in a process group of ONE rank (`NEMO_BENCH_SHARD_OF_ONE=1`: ShardedNemo, RCCL communicator of world size 1) "
         "-- what the sharded launch structure costs by itself.  Since the last third of round 3 a sharded step is ONE captured launch: the "
         "RCCL all-reduce(s), the loss hand-over and the fused Adam sit inside the step's HIP graph; `NEMO_GRAPH_COMM=0` = the earlier "
         "structure (graph, then eager all-reduce / hand-over / Adam), same box\n\n"
         "| instances | structure | it/s | ms/step | modes timed by `--shard-mode auto` (ms/step) | kept | compute_ms (collectives skipped) | collective_ms (9 MB all-reduce alone, world of one) |\n"
         "|---:|---|---:|---:|---|---|---:|---:|\n")
for v in (1, 2, 4):
    for fn, lab in ((f'bench_group1_v{v}.json', 'collectives in the graph'), (f'bench_group1_eager_v{v}.json', 'NEMO_GRAPH_COMM=0')):
        if not os.path.exists(os.path.join(G, fn)):
            continue
        d = line(fn)
        pr = d['per_rank'][0]
        txt += (f"| {v} | {lab} | {d['value']} | {d['ms_per_step']} | {json.dumps(d['shard_modes_ms'])} | {d['shard_mode']} | {pr['compute_ms']} | "
                f"{pr['collective_ms']} |\n")
txt += ("\n`buckets` (three gradient buckets reduced behind the backward) loses on one GPU at these sizes even as one launch: its backward "
        "runs the parameter gradients on the critical chain (a bucket must be complete before its collective) and a world of one gains "
        "nothing from the overlap -- 0.66 - 0.67 ms at one instance against 0.51 - 0.52 for `split` when all three were timed (`r03_experiments.md` section 16); `--shard-mode auto` times `buckets` only for per-rank steps of >= 4096 samples, so it is not in the lines above.\n\n"
        "## the full default line (what the driver records)\n```\n" + json.dumps(full) + "\n```\n"
        f"cpu_baseline: {json.dumps(full['cpu_baseline'])}\n")
if os.path.exists(os.path.join(G, 'bf16mem_gemm.txt')):
    txt += ('\n## nemo_gemm_bf16mem against nemo_gemm_bf16 / nemo_gemm_f32 at the C3 shapes (tools/bench_bf16mem.py; HIP events, 20 launches)\n```\n'
            + '\n'.join(l for l in rd('bf16mem_gemm.txt').splitlines() if 'GFLOP' in l) + '\n```\n')
open(os.path.join(P, 'r03_bench_lines.md'), 'w').write(txt)


# ---- the shard budget: un-profiled contribution of single kernels to the step (NEMO_ABLATE)
def ab(fn):
    p = os.path.join(G, fn) if os.path.exists(os.path.join(G, fn)) else os.path.join(GO, fn)
    return [l for l in open(p).read().splitlines() if l.startswith(('full step', 'nemo_')) and 'ms (delta' in l or l.startswith('full step')]


sb = (f"# Round 3 (commit {head}) -- where the time of a SHARD-sized step goes, un-profiled\n\n"
      "Method: `bash tools/ablate.sh <instances> <entry point> ...` = `bench.py --instances k --steps 100` with ONE C entry point of "
      "libnemo_hip.so turned into a no-op (`NEMO_ABLATE`, nemo_cvpr2023_amd/_lib.py; `nemo_gemm_f32@MxNxK` = one GEMM shape).  The step then "
      "computes garbage; the DIFFERENCE of its time to the full step is what the kernel(s) contribute to the critical path of the replayed "
      "graph WITHOUT a profiler attached (kernel traces over-state cross-queue latencies inside replayed graphs, DESIGN.md 5a: e.g. the "
      "traces of r03_kernel_trace_v1.md show ~600 us per step where the un-profiled step takes 0.50 ms).  A delta near zero = the kernel is "
      "off the critical path (another branch of the graph is longer); a delta above the kernel's traced duration = it also carries a "
      "fork / join of the graph.\n\n"
      "Shard sizes of the 8 x 300 problem: 1 instance = one rank of 8 GPUs (N = 300), 2 = one rank of 4 (N = 600), 4 = one rank of 2 "
      "(N = 1200), 8 = the single-GPU step (N = 2400).\n\n")
names = {'ablate_v1.txt': 'N = 300 (1 instance)', 'ablate_v1b.txt': 'N = 300, the remaining GEMM shapes', 'ablate_v2.txt': 'N = 600 (2 instances)',
         'ablate_v4.txt': 'N = 1200 (4 instances)', 'ablate_v8.txt': 'N = 2400 (8 instances: the headline step)'}
for fn, title in names.items():
    try:
        sb += f"## {title}\n```\n" + '\n'.join(ab(fn)) + "\n```\n\n"
    except OSError:
        pass
sb += open(os.path.join(ROOT, 'profiles', 'r03_shard_budget_notes.md')).read() if os.path.exists(os.path.join(ROOT, 'profiles', 'r03_shard_budget_notes.md')) else ''
open(os.path.join(P, 'r03_shard_budget.md'), 'w').write(sb)
print('wrote profiles for', head, '; step traffic MiB', round(tot))
