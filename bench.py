#!/usr/bin/env python3
"""NeMo fit benchmark: iterations/second of the per-iteration optimisation step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (``WORLD_SIZE`` unset) this process starts N ranks itself --
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` as a CHILD
process, before anything here touches the GPU -- and exits with the children's code.  Launched by
``torch.distributed.run`` directly (the driver's way) the ranks run as they are.  Either way every rank checks
that the world it sees is the world it was asked for (backend ``nccl`` = RCCL, ``world_size == --gpus``, one
distinct device per rank) and rank 0 prints ``ranks_seen``; fewer visible GPUs than ``--gpus`` is an error
(exit code 3), never a silent single-GPU number.

Workload (config.workload): the Baseball-Pitch fit of BASELINE.json configs[1] -- 8 instances x 300
frames, full batch N = 2400 (instance x frame) samples per step, NemoV2 with the published-run
hyper-parameters (h_dim 1000, RBF 100, instance code 5, mse_robust, w_vp 10, w_vp_z 1, w_gmm 1; every
loss term evaluated as the reference does), 6890-vertex SMPL, fp32, synthetic SMPL-shaped assets and
random OpenPose targets (no dataset / licensed model files on the box).  One "step" = one
``model.step(update=True)``: forward, backward, Adam on all four optimisers, loss read-back.

N > 1: strong scaling -- the same 8 x 300 problem sharded by instance over the ranks
(nemo_cvpr2023_amd/dist.py).  The library default is ONE RCCL all-reduce per step (shared MLP gradient + loss
scalars in one buffer); ``--shard-mode auto`` (the bench default for N > 1) times that against the two-collective
variant (loss scalars reduced early on a side stream, see dist.py) for a few steps on the actual machine, takes
the faster one on all ranks and reports both timings and ``collectives_per_step``.

Output (rank 0, ONE JSON line): metric/value/... per the driver contract plus
  roofline     -- the dominant kernel, timed live with HIP events on its launch stream (in a short
                  instrumented pass of the same steps: the timed region replays a captured HIP graph);
                  achieved = algorithmic FLOPs per launch / mean launch time vs the MFMA peak of the dtype;
                  `step` = the whole step's algorithmic FLOPs / step time; `hbm` = counter bytes per step
                  (profiles/) / step time vs 8 TB/s
  cpu_baseline -- the CPU oracle ("port" of the reference PyTorch path) timed on this box's host cores:
                  SURVEY 8(d) protocol, 3 warm-up + 10 timed steps at C2 full batch (N = 2400), C2 minibatch
                  (N = 512) and C1 (1 x 30, default-v1), each in a child process with a wall-clock bound
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V0, T0 = 8, 300
# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks (fp32: v_mfma_f32_32x32x2_f32; bf16: 32x32x16) and HBM3E
MFMA_PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'traffic.json')


def step_flops(n, nv=6890, h=1000, din=105):
    """Algorithmic FLOPs of one published-configuration update step as THIS engine formulates it (DESIGN.md
    section 4): fused mesh term (2 pose blends, 2 skinnings, vertex->joint adjoint), blend-shape adjoint, MLP
    forward + activation-gradient + parameter-gradient GEMMs, VPoser encode/decode + KL backward, the
    pre-contracted joint GEMM and its adjoint, GMM prior.  Returns (total, per-part dict)."""
    mesh = 2.0 * n * nv * (2 * 3 * 207 + 2 * 288 + 288)
    adj = 2.0 * n * 207 * 3 * nv
    mlp = 3 * 2.0 * (n + 1) * (din * h + 2 * h * h + h * 147)
    vposer = 2.0 * n * (63 * 512 + 512 * 64 + 32 * 512 + 512 * 512 + 512 * 126) + 2.0 * n * (64 * 512 + 512 * 63)
    joints = 2 * 2.0 * n * 207 * 792
    gmm = 2.0 * n * 8 * 69 * 70
    parts = dict(mesh=mesh, blend_adjoint=adj, mlp=mlp, vposer=vposer, joints=joints, gmm=gmm)
    return sum(parts.values()), parts


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=10)
    ap.add_argument('--cpu-warmup', type=int, default=3)
    ap.add_argument('--cpu-timeout', type=float, default=150.0, help='wall-clock bound of one CPU-baseline child')
    ap.add_argument('--no-torch-gpu-baseline', action='store_true')
    ap.add_argument('--instances', type=int, default=V0, help='exploration only; the judged workload is 8')
    ap.add_argument('--frames', type=int, default=T0, help='exploration only; the judged workload is 300')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="GEMM operand type; 'bf16' = BASELINE configs[2] (fp32 accumulate, fp32 master weights)")
    ap.add_argument('--shard-mode', choices=['auto', 'single', 'split', 'buckets'], default='auto',
                    help='N > 1: how the shared gradient is reduced (single = one collective, the library default; split = + an '
                         'early 32-byte one for the losses; buckets = three buckets behind the backward; auto = time all, keep '
                         'the fastest)')
    ap.add_argument('--repeat', type=int, default=5,
                    help='timed regions of --steps steps each; `value` is the MEDIAN region (a region lasts ~30 - 50 ms at the '
                         'default --steps: one region alone is at the mercy of a single scheduling hiccup)')
    ap.add_argument('--minibatch-steps', type=int, default=60,
                    help='extra, separately reported leg: random minibatches of 512 samples (the published run\'s mode); 0 = skip')
    ap.add_argument('--spawn-selftest', action='store_true',
                    help='test aid: the N ranks only rendezvous (gloo, no GPU) and report ranks_seen')
    ap.add_argument('--cpu-child', default='', help=argparse.SUPPRESS)
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- launcher
def spawn_ranks(opts):
    """Parent of an N-rank run.  Touches no GPU: ``device_count()`` does not initialise HIP on this image."""
    if not opts.spawn_selftest:
        import torch
        have = torch.cuda.device_count()
        if have < opts.gpus:
            sys.stderr.write(f'bench.py: --gpus {opts.gpus} but only {have} GPU(s) visible on this node; refusing to '
                             f'print a {have}-GPU number for a {opts.gpus}-GPU request\n')
            return 3
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // opts.gpus)))
    if opts.spawn_selftest:
        env['NEMO_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={opts.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def check_world(opts, dist, torch, rank, world, local_rank):
    """Every rank: the process group is what --gpus asked for.  Returns ranks_seen (sorted rank ids)."""
    want = os.environ.get('NEMO_DIST_BACKEND', 'nccl')
    if dist.get_world_size() != opts.gpus or dist.get_backend() != want:
        raise SystemExit(f'bench.py: process group is {dist.get_backend()} x {dist.get_world_size()}, '
                         f'asked for {want} x {opts.gpus}')
    seen = [None] * world
    dev = None if opts.spawn_selftest else torch.cuda.current_device()
    dist.all_gather_object(seen, (rank, local_rank, dev))
    ranks = sorted(r for r, _, _ in seen)
    if ranks != list(range(opts.gpus)):
        raise SystemExit(f'bench.py: ranks seen {ranks}, expected 0..{opts.gpus - 1}')
    if want == 'nccl' and len({d for _, _, d in seen}) != world:
        raise SystemExit(f'bench.py: {world} ranks share devices {[d for _, _, d in seen]}')
    return ranks


# ----------------------------------------------------------------------------------------------- CPU leg
def cpu_child(spec):
    """``--cpu-child name,V,T,B,threads,warmup,steps``: time the CPU oracle (child process, no GPU)."""
    name, V, T, B, threads, warm, steps, *dev = spec.split(',')
    V, T, B, threads, warm, steps = int(V), int(T), int(B), int(threads), int(warm), int(steps)
    dev = dev[0] if dev else 'cpu'            # 'cuda': the same unfused step through PyTorch-ROCm (torch_gpu_baseline)
    import torch
    from nemo_cvpr2023_amd import synthetic as syn
    from oracle.model import OracleNemo
    if threads > 0:
        torch.set_num_threads(threads)
    if name == 'c1':
        version, args, nv = 1, syn.default_v1_args(batch_size=-1, out_dir=''), 6890
    else:
        version, args, nv = 2, syn.published_args(batch_size=B if B > 0 else 512, out_dir=''), 6890
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    torch.manual_seed(0)
    o = OracleNemo(version, args, seqs, syn.make_smpl_assets(nv, seed=1), syn.make_vposer_state(), syn.make_gmm())
    gen = torch.Generator().manual_seed(2)
    sync = (lambda: None) if dev == 'cpu' else torch.cuda.synchronize
    if dev != 'cpu':
        o = o.to(dev)
        torch.set_default_device(dev)

    def one():
        if B > 0:
            o.step(torch.randint(0, V, (B,), generator=gen), torch.randint(0, T, (B,), generator=gen), update=True)
        else:
            o.step(None, None, update=True, full_batch=True)
    for _ in range(warm):
        one()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    sync()
    dt = time.perf_counter() - t0
    print(json.dumps({'value': round(steps / dt, 4), 'seconds': round(dt, 2)}))


def cpu_baseline(opts):
    """the oracle (a port of the reference step) on the host cores, 3 warm-up + 10 timed steps of C2
    full batch (N = 2400; the headline `value`), C2 minibatch (N = 512) and C1 (1 x 30, default-v1).  PyTorch's
    CPU backend collapses when its thread pool is far larger than the op sizes warrant (round 1: 196 s/step with
    256 threads), so the headline uses min(cores, 16) threads and the ``os.cpu_count()`` figure is reported next to
    it from a wall-clock-bounded probe (1 warm-up + 2 steps of the N = 512 case)."""
    ncpu = os.cpu_count() or 1
    thr = min(ncpu, 16)

    def run(name, V, T, B, threads, warm, steps, timeout):
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-child', f'{name},{V},{T},{B},{threads},{warm},{steps}']
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            return json.loads(line[-1]) if line else {'error': (r.stderr or 'no output')[-200:]}
        except subprocess.TimeoutExpired:
            return {'error': f'timeout > {timeout:.0f} s'}
    w, k = opts.cpu_warmup, opts.cpu_steps
    full = run('c2', V0, T0, -1, thr, w, k, opts.cpu_timeout)
    mini = run('c2', V0, T0, 512, thr, w, k, opts.cpu_timeout)
    c1 = run('c1', 1, 30, -1, thr, w, k, opts.cpu_timeout)
    allc = run('c2', V0, T0, 512, ncpu, 1, 2, 60.0) if ncpu != thr else mini
    return {'value': full.get('value'), 'unit': 'iters/s', 'cores': thr, 'kind': 'port',
            'sample': f'{k} full-batch steps ({V0}x{T0}, N={V0 * T0}) of the CPU oracle (plain PyTorch fp32 restatement '
                      f'of the reference step) after {w} warm-up steps, {thr} threads, {full.get("seconds")} s',
            'c2_minibatch512': dict(mini, cores=thr), 'c1_1x30_default_v1': dict(c1, cores=thr),
            'all_cores_probe': dict(allc, cores=ncpu, sample='N=512 minibatch, 1 warm-up + 2 steps, 60 s bound'),
            'cpu_count': ncpu, **({'error': full['error']} if 'error' in full else {})}


# ----------------------------------------------------------------------------------------------- main
def main():
    opts = parse()
    if opts.cpu_child:
        return cpu_child(opts.cpu_child)
    if opts.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and opts.gpus > 1:
        sys.exit(spawn_ranks(opts))
    V, T = opts.instances, opts.frames
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != opts.gpus:
        raise SystemExit(f'bench.py: --gpus {opts.gpus} but WORLD_SIZE={world}')

    import torch
    import torch.distributed as dist

    if opts.spawn_selftest:                                   # rendezvous only (CPU test of the launcher)
        dist.init_process_group('gloo')
        ranks = check_world(opts, dist, torch, rank, world, local_rank)
        if rank == 0:
            print(json.dumps({'selftest': True, 'n_gpus': world, 'ranks_seen': ranks}))
        dist.destroy_process_group()
        return

    from nemo_cvpr2023_amd import synthetic as syn
    have = torch.cuda.device_count()
    backend = os.environ.get('NEMO_DIST_BACKEND', 'nccl')
    if have < (world if backend == 'nccl' else 1):
        raise SystemExit(f'bench.py: {world} ranks but {have} GPU(s) visible')
    device = f'cuda:{local_rank % have}'
    torch.cuda.set_device(device)
    ranks_seen = [0]
    # diagnostic: NEMO_BENCH_SHARD_OF_ONE=1 runs the SHARDED code path (ShardedNemo, the all-reduce, Adam after it, both
    # shard modes) in a process group of one rank -- what the sharded host / launch structure costs by itself on one GPU
    sharded = world > 1
    if world == 1 and os.environ.get('NEMO_BENCH_SHARD_OF_ONE') == '1':
        sharded = True
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend, rank=0, world_size=1)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL on ROCm.  (NEMO_DIST_BACKEND=gloo: test aid for boxes where several ranks must share one GPU,
        # which RCCL refuses.)
        dist.init_process_group(backend)
        ranks_seen = check_world(opts, dist, torch, rank, world, local_rank)

    args = syn.published_args(batch_size=512, out_dir='')
    args.gemm_dtype = opts.dtype
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets = dict(smpl_assets=syn.make_smpl_assets(6890, seed=1), vposer_state=syn.make_vposer_state(),
                  gmm=syn.make_gmm())
    torch.manual_seed(0)
    if sharded:
        from nemo_cvpr2023_amd.dist import ShardedNemo
        model = ShardedNemo(2, args, seqs, device, rank=rank, world=world, seed=0, **assets)
        engine = model.model.engine
    else:
        from nemo_cvpr2023_amd.neural_motion_model import NemoV2
        model = NemoV2(args, seqs, device, **assets)
        engine = model.engine

    def step():
        return model.step(None, None, update=True, full_batch=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt, out

    # set-up: the first calls of a (batch size, mode) variant allocate its workspace and capture its HIP graph
    # (like a compile step); they are not part of the warm-up / timed protocol below
    shard_modes = None
    if sharded:
        # auto: one collective per step against the early loss all-reduce against three gradient buckets behind the backward.
        # With RCCL every mode is ONE captured launch per step (collectives inside the graph); where they are not capturable
        # (gloo) the three-bucket layout costs two more launches per step and is only tried for long per-rank steps
        # -- and in any case only for long per-rank steps: in a group of one it needs > 150 us of hidden collective per step to
        # catch up with `split` (0.67 against 0.52 ms at one instance), more than the whole 9 MB all-reduce takes
        modes = (['single', 'split'] + (['buckets'] if V * T // max(world, 1) >= 4096 else [])) \
            if opts.shard_mode == 'auto' else [opts.shard_mode]
        shard_modes = {}
        for mode in modes:
            model.set_shard_mode(mode)
            for _ in range(4):
                step()
            shard_modes[mode] = round(1e3 * timed(10)[0] / 10, 4)
        best = min(shard_modes, key=shard_modes.get)       # max-over-ranks timings: the same choice on every rank
        model.set_shard_mode(best)
    for _ in range(3):
        step()
    for _ in range(opts.warmup):
        step()
    # EXACTLY --steps steps per timed region (barrier + synchronize on both sides, max over ranks); --repeat regions, the
    # median one is reported (all of them are listed in `repeat_ms_per_step`)
    regions = []
    for _ in range(max(1, opts.repeat)):
        regions.append(timed(opts.steps))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, (ld, _) = regions[order[len(order) // 2]]
    repeat_ms = [round(1e3 * r[0] / opts.steps, 4) for r in regions]

    # ---- sharded runs: what one rank computes per step without any collective, and what the collective costs alone -- so
    # that a multi-GPU number can be read as  step = compute + (un-hidden part of the) collective
    shard_info = None
    if sharded:
        n_probe = max(10, min(50, opts.steps))
        try:
            model.collectives = False                             # the sharded code path with the collectives skipped
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_probe):
                step()
            torch.cuda.synchronize()
            compute_ms = 1e3 * (time.perf_counter() - t0) / n_probe
        finally:
            model.collectives = True
        a_, b_ = model._span
        buf = torch.zeros(b_ - a_, device=device)
        for _ in range(5):
            dist.all_reduce(buf)
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        coll_ms = 1e3 * (time.perf_counter() - t0) / 20
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {'rank': rank, 'compute_ms': round(compute_ms, 4), 'collective_ms': round(coll_ms, 4),
                                          'instances': model.plan.v_local, 'device': torch.cuda.current_device()})
        worst = max(p_['compute_ms'] for p_ in per_rank)
        shard_info = {'per_rank': per_rank, 'allreduce_bytes': int((b_ - a_) * 4),
                      'scaling_model': {'shard_step_ms': worst, 'allreduce_ms': max(p_['collective_ms'] for p_ in per_rank),
                                        'predicted_ms_per_step_no_overlap': round(worst + max(p_['collective_ms'] for p_ in per_rank), 4),
                                        'note': 'compute_ms = this rank\'s step with every collective stubbed out; collective_ms = '
                                                'the all-reduce of the shared-gradient slice alone (back-to-back, synchronised); '
                                                'the measured ms_per_step lies between max(compute) and compute + collective'}}

    # ---- the published run's mode: random minibatches of 512 samples drawn like the script does (CPU RNG, views then frames)
    mini = None
    if opts.minibatch_steps > 0 and V * T >= 512:
        gen = torch.Generator().manual_seed(1234)
        draws = [(torch.randint(0, V, (512,), generator=gen), torch.randint(0, T, (512,), generator=gen))
                 for _ in range(opts.minibatch_steps + 12)]
        mm = model.model if sharded else model
        for vi_, fi_ in draws[:12]:                     # set-up: workspaces + graph captures of the launch sizes
            model.step(vi_, fi_)
        stats0 = dict(mm.launch_stats)
        barrier()
        t0 = time.perf_counter()
        for vi_, fi_ in draws[12:]:
            model.step(vi_, fi_)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        rep = mm.launch_stats['replayed'] - stats0['replayed']
        oth = mm.launch_stats['other'] - stats0['other']
        mini = {'value': round(opts.minibatch_steps / dt, 2), 'unit': 'iters/s', 'ms_per_step': round(1e3 * dt / opts.minibatch_steps, 4),
                'batch': 512, 'steps': opts.minibatch_steps, 'graph_replay_fraction': round(rep / max(rep + oth, 1), 3),
                'note': 'random (view, frame) minibatches of 512 out of %d x %d, drawn on the host like '
                        'scripts/learned_multi_view_recon_nn.py:291-296; rank 0\'s launches' % (V, T)}
    # Roofline leg: the production step replays a captured HIP graph (events cannot be recorded inside
    # one), so the same steps are run once more, un-captured, with HIP events around the tagged kernels
    # on their launch streams.  These instrumented steps are not part of `value`.
    engine.timers = {}
    n_inst = max(3, min(10, opts.steps))
    for _ in range(n_inst):
        step()
    barrier()
    timers, engine.timers = engine.timers, None
    ms_per_step = 1e3 * elapsed / opts.steps
    iters_per_s = opts.steps / elapsed
    peak = MFMA_PEAK_TFLOPS[opts.dtype]

    # dominant tagged kernel (largest total time in the timed region)
    roof = None
    best = None
    for tag, evs in timers.items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        tot = sum(ms)
        if best is None or tot > best[0]:       # mean launch time and mean FLOPs per launch (chunked batches differ)
            best = (tot, tag, sum(ms) / len(ms), sum(e[2] for e in evs) / len(evs), len(ms))
    if best is not None:
        _, tag, mean_ms, flops, n = best
        # Price every part of the kernel against the pipe it runs on: with --dtype bf16 only the pose blend of the fused
        # mesh kernel is bf16 work, its skinning / L1 stay on the fp32 pipe (the vertex->joint adjoint: bf16 pipe, split precision).  The peak quoted is the rate at
        # which the kernel's own mix of work would run with both pipes at their peaks (harmonic mix; = the fp32 peak
        # for the fp32 build).
        f_step, parts = step_flops(V * T // world if world > 1 else V * T)
        pipes = engine.kernel_flops_by_pipe(tag, flops)
        kpeak = flops / sum(f / MFMA_PEAK_TFLOPS[d] for d, f in pipes.items())
        achieved = flops / (mean_ms * 1e-3) / 1e12
        traffic = {}
        if os.path.exists(TRAFFIC_FILE):
            traffic = json.load(open(TRAFFIC_FILE)).get(f'{V}x{T}x{world}x{opts.dtype}', {})
        ktr = traffic.get('kernels', {}).get(tag)
        # the whole step, same pricing: which parts of step_flops() the bf16 build moves to the bf16 pipe
        mesh_b16 = engine.kernel_flops_by_pipe('mesh_v2v_fused', parts['mesh']).get('bf16', 0.0)
        on_bf16 = (mesh_b16 + parts['blend_adjoint'] + parts['mlp'] + parts['vposer']) if opts.dtype == 'bf16' else 0.0
        step_peak = f_step / (on_bf16 / MFMA_PEAK_TFLOPS['bf16'] + (f_step - on_bf16) / MFMA_PEAK_TFLOPS['f32'])
        roof = {'kernel': tag, 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(kpeak, 1),
                'unit': 'TFLOP/s', 'frac': round(achieved / kpeak, 4),
                'pipes': {d: round(f / 1e9, 2) for d, f in pipes.items()},
                'peak_note': 'fp32 MFMA peak' if len(pipes) == 1 and 'f32' in pipes else
                             'harmonic mix of the bf16 (2500) and fp32 (157.3) MFMA peaks over this kernel\'s GFLOP per pipe '
                             '(`pipes`); the fp32 skinning part bounds it',
                'traffic': ktr, 'traffic_source': traffic.get('source') if ktr else None,
                'launches': n, 'mean_launch_ms': round(mean_ms, 4), 'flops_per_launch': flops,
                'timed_in': f'{n_inst} instrumented (un-captured) steps after the timed region',
                'per_kernel_ms_per_step': {t: round(sum(a.elapsed_time(b) for a, b, _ in e) / n_inst, 4)
                                           for t, e in timers.items()},
                # the whole step against both roofs: algorithmic FLOPs of one rank's step / step time against the peak of
                # its own mix of pipes (fp32 build: the fp32 MFMA peak), and counter-measured HBM-side bytes per step
                # (separate --pmc passes, profiles/) / step time
                'step': {'flops': f_step, 'achieved': round(f_step / (ms_per_step * 1e-3) / 1e12, 2),
                         'peak': round(step_peak, 1), 'unit': 'TFLOP/s',
                         'frac': round(f_step / (ms_per_step * 1e-3) / 1e12 / step_peak, 4),
                         'gflop_on_bf16_pipe': round(on_bf16 / 1e9, 2),
                         'flops_parts': {k: round(v / 1e9, 2) for k, v in parts.items()}},
                'hbm': None}
        if traffic.get('step_bytes'):
            gbs = traffic['step_bytes'] / (ms_per_step * 1e-3) / 1e9
            roof['hbm'] = {'bytes_per_step': traffic['step_bytes'], 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                           'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'source': traffic.get('source')}
        roof['hbm_frac'] = roof['hbm']['frac'] if roof['hbm'] else None

    cpu = None
    if rank == 0 and world == 1 and not opts.no_cpu_baseline:
        cpu = cpu_baseline(opts)

    # The reference's own formulation (unfused PyTorch autograd, = the oracle) through PyTorch-ROCm's stock
    # kernels on THIS GPU: what `north_star` calls "the reference single-GPU PyTorch iters/sec".
    tgpu = None
    if rank == 0 and world == 1 and not opts.no_torch_gpu_baseline:
        # in a child process: the stock kernels fault on this GPU from ~24 x 300 samples on (32-bit indexing of the
        # (N, 6890, 24, ...) skinning intermediates), and a baseline must never take the bench line down
        n_t = 5
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-child', f'c2,{V},{T},-1,0,2,{n_t},cuda']
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            gval = json.loads(r.stdout.strip().splitlines()[-1])['value']
            tgpu = {'value': round(gval, 3), 'unit': 'iters/s', 'kind': 'port',
                    'sample': f'{n_t} full-batch steps ({V}x{T}) of the oracle (unfused PyTorch restatement of the '
                              f'reference step) on this GPU through PyTorch-ROCm (child process), after 2 warm-up steps',
                    'speedup_of_this_engine': round(iters_per_s / gval, 1)}
        except Exception as exc:
            tail = ''
            try:
                tail = (r.stderr or '').strip().splitlines()[-1][:160]
            except Exception:
                pass
            tgpu = {'error': (repr(exc)[:120] + ' | ' + tail).strip(' |')}

    if rank == 0:
        out = {
            'metric': 'NeMo fit iters/sec (instances x frames/step), Baseball-Pitch',
            'value': round(iters_per_s, 3), 'unit': 'iters/s', 'n_gpus': world, 'steps': opts.steps,
            'warmup': opts.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': opts.dtype, 'data': 'synthetic',
            'samples_per_s': round(iters_per_s * V * T, 1),
            'config': {'workload': f'Baseball-Pitch-shaped fit, {V} instances x {T} frames full batch (N={V * T}), '
                                   'NemoV2 published hyper-parameters, all loss terms, 6890-vertex SMPL',
                       'instances': V, 'frames': T, 'samples_per_step': V * T, 'h_dim': args.h_dim,
                       'parallelism': f'instance-shard x{world}' if world > 1 else
                       ('sharded code path in a process group of ONE rank (diagnostic)' if sharded else 'single GPU')},
            'ranks_seen': ranks_seen,
            'repeat': len(repeat_ms), 'repeat_ms_per_step': repeat_ms,
            'minibatch512': mini,
            'final_total_loss': float(ld['total_loss']),
            'roofline': roof, 'cpu_baseline': cpu, 'torch_gpu_baseline': tgpu,
        }
        if sharded:
            out['backend'] = dist.get_backend()
            out['shard_modes_ms'] = shard_modes
            out['shard_mode'] = model.shard_mode
            out['collectives_per_step'] = {'single': 1, 'split': 2, 'buckets': 3}[model.shard_mode]
            out.update(shard_info)
        print(json.dumps(out))
    if sharded:
        model.close()                  # captured graphs hold RCCL launches: released before the communicator
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
