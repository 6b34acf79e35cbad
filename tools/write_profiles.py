#!/usr/bin/env python3
"""Turns the files a profiling gpurun call leaves under gpurun_out/ into the committed summaries
profiles/r01b_kernel_trace.md, r01b_pmc_traffic.md, r01c_gemm_sweep.md, and refreshes bench.py's PMC_TRAFFIC
constant.  usage: write_profiles.py <steps traced by the kernel-trace run>"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
run = lambda *a: subprocess.run([sys.executable] + list(a), capture_output=True, text=True, cwd=ROOT).stdout
head = subprocess.run(['git', 'log', '--oneline', '-1'], capture_output=True, text=True, cwd=ROOT).stdout.split()[0]
txt = lambda f: ''.join(l for l in open(os.path.join(G, f)) if 'amdgpu.ids' not in l)
steps = sys.argv[1] if len(sys.argv) > 1 else '35'

prof = json.loads([l for l in open(os.path.join(G, 'prof_b.log')) if l.startswith('{')][-1])
full = json.loads(open(os.path.join(G, 'bench_full.json')).read().strip().split('\n')[-1])
r = prof['roofline']
open(os.path.join(ROOT, 'profiles', 'r01b_kernel_trace.md'), 'w').write(
    f"# Round 1 (final snapshot, commit {head}) -- rocprofv3 --kernel-trace --stats of bench.py ({steps} steps traced: "
    f"set-up + warm-up + timed graph replays + 10 instrumented eager steps)\n\n"
    "Command (on the MI355X box): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats "
    "-d gpurun_out/prof_trace -o r01b -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-torch-gpu-baseline`\n\n"
    f"bench line of the same (profiled) run: {prof['value']} it/s, {prof['ms_per_step']} ms/step; roofline kernel {r['kernel']} "
    f"{r['mean_launch_ms']} ms/launch (HIP events in bench.py) -> {r['achieved']} TFLOP/s\n"
    f"(un-profiled default run of the same commit: {full['value']} it/s, {full['ms_per_step']} ms/step, {r['kernel']} "
    f"{full['roofline']['mean_launch_ms']} ms/launch -> {full['roofline']['achieved']} TFLOP/s; cpu_baseline "
    f"{full['cpu_baseline']['value']} it/s on {full['cpu_baseline']['cores']} threads; torch_gpu_baseline "
    f"{full['torch_gpu_baseline']['value']} it/s)\n\n"
    + run('tools/prof_summary.py', os.path.join(G, 'prof_trace', 'r01b_results.db'), steps))
pmc = run('tools/pmc_summary.py', os.path.join(G, 'pmc_fetch', 'f_results.db'), os.path.join(G, 'pmc_write', 'w_results.db'))
open(os.path.join(ROOT, 'profiles', 'r01b_pmc_traffic.md'), 'w').write(
    f"# Round 1 (final snapshot, commit {head}) -- HBM-side traffic, separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)\n\n"
    "Commands: `export NEMO_GRAPHS=0; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py --steps 4 --warmup 1 "
    "--no-cpu-baseline --no-torch-gpu-baseline` and the same with `--pmc WRITE_SIZE` (eager launches so that every kernel is its "
    "own dispatch).\nFETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 column is the "
    "corrected read volume for wide coalesced streams. Infinity-Cache hits are counted.\n\n"
    "mesh_v2v_fused: compulsory output = dVP^T 3*6896*2400*4 B = 189 MiB + 11 MiB of per-range dA partials; inputs (pose "
    "features, transforms, W, v_shaped, 17 MB of blend shapes) < 30 MiB when cached.  Measured below: the write side is the "
    "compulsory output (+ scratch of the 12 registers the epilogue spills); the read side is the blend-shape matrix re-fetched by "
    "the 8 XCD L2s (it does not fit a 4 MB L2; the reads hit the Infinity Cache).\n"
    "gemm <64,64,32,true,true,...> = the blend-shape adjoint dPF = (dVP^T)^T P^T (K = 20670): reads dVP^T once per 64-column "
    "tile of the 207 outputs (4 x 189 MiB).\n\n" + pmc)
open(os.path.join(ROOT, 'profiles', 'r01c_gemm_sweep.md'), 'w').write(
    f"# Round 1 -- nemo_gemm_f32 (tile, split-K) sweep on one MI355X, the fp32 MFMA peak of the box, and the vendor fp32 GEMM on "
    f"the step's shapes (commit {head})\n\n"
    "Commands: `python3 tools/bench_gemm.py mlp,rot_out,vposer,mq,dpf,pose_blend_bwd --sweep 20`, `python3 tools/bench_gemm.py "
    "mlp,rot_out,vposer 20 300 --sweep`, `./tools/mfma_peak 2000`, `python3 tools/bench_torch_mm.py` (HIP events around 20 "
    "back-to-back launches; microseconds per launch incl. the ~4.5 us dispatch gap).\n`auto` = the plan of the host cost model in "
    "gemm.hip; `sN` = forced N K-slices combined in the launch by the last-arriving block.\n\n"
    "## step shapes at N = 2400 (one GPU)\n```\n" + txt('gemm_sweep_2400.txt') + "```\n"
    "## step shapes at N = 300 (one rank of 8)\n```\n" + txt('gemm_sweep_300.txt') + "```\n"
    "## vendor fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the same shapes\n```\n" + txt('torch_mm.txt') + "```\n"
    "## fp32 MFMA issue rate (tools/mfma_peak.hip)\n```\n" + txt('mfma_peak.txt') + "```\n")
if os.path.exists(os.path.join(G, 'pmc_mfma', 'm_counter_collection.csv')):
    open(os.path.join(ROOT, 'profiles', 'r01e_pmc_mfma.md'), 'w').write(
        f"# Round 1 (commit {head}) -- MFMA-pipe and LDS counters per kernel, separate rocprofv3 --pmc passes\n\n"
        "Commands (`tools/profile_mfma.sh`, eager launches): `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
        "SQ_INSTS_VALU_MFMA_MOPS_F32 ... -- python3 bench.py --steps 4 --warmup 1 ...` and `--pmc SQ_LDS_BANK_CONFLICT "
        "SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES`.\nMFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); "
        "executed GFLOP = MOPS_F32 x 512 (the mesh kernel: 73.1 executed for 69.65 algorithmic -- the half-empty second joint "
        "tile of the adjoint).\nThese counters found: 576 executed MFMAs per mesh tile where 552 suffice (four zero-padding k-steps), "
        "a 16-way bank conflict in the mesh kernel's cross-wave dA reduction (288-float stride), 2-way conflicts of the GEMM's "
        "transposing LDS stores (lane mapping per half-wave) -- all fixed in the numbers below.\n\n"
        + run('tools/pmc_mfma_summary.py'))
for l in pmc.split('\n'):
    if 'mesh_v2v_fused' in l:
        c = [x.strip() for x in l.split('|')]
        s = open(os.path.join(ROOT, 'bench.py')).read()
        s = re.sub(r"int\(\(\d+\.\d+ \+ \d+\.\d+\) \* 2 \*\* 20\)", f"int(({float(c[4])} + {float(c[5])}) * 2 ** 20)", s)
        open(os.path.join(ROOT, 'bench.py'), 'w').write(s)
        print('mesh traffic MiB (x2 fetch, write):', c[4], c[5])
print('wrote profiles for', head)
