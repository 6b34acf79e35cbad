#!/usr/bin/env python3
"""NeMo fit benchmark: iterations/second of the per-iteration optimisation step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): the Baseball-Pitch fit of BASELINE.json configs[1] -- 8 instances x 300
frames, full batch N = 2400 (instance x frame) samples per step, NemoV2 with the published-run
hyper-parameters (h_dim 1000, RBF 100, instance code 5, mse_robust, w_vp 10, w_vp_z 1, w_gmm 1; every
loss term evaluated as the reference does), 6890-vertex SMPL, fp32, synthetic SMPL-shaped assets and
random OpenPose targets (no dataset / licensed model files on the box).  One "step" = one
``model.step(update=True)``: forward, backward, Adam on all four optimisers, loss read-back.

Output (rank 0, ONE JSON line): metric/value/... per the driver contract plus
  roofline      -- the dominant kernel, timed live with HIP events on its launch stream (in a short
                   instrumented pass of the same steps: the timed region replays a captured HIP graph);
                   achieved = algorithmic FLOPs per launch / mean launch time vs the MFMA peak of the dtype;
                   `step` = the whole step's algorithmic FLOPs / step time; `hbm` = counter bytes per step
                   (profiles/) / step time vs 8 TB/s
  cpu_baseline  -- the CPU oracle ("port" of the reference PyTorch path) timed on this box's host cores:
                   SURVEY 8(d) protocol, 3 warm-up + 10 timed steps at C2 full batch (N = 2400), C2 minibatch
                   (N = 512) and C1 (1 x 30, default-v1), each in a child process with a wall-clock bound
  minibatch512  -- random minibatches of 512 (the published run's mode)
  published_fit -- wall clock of the published schedule (300 warm-up + 1000 camera + 2000 minibatch-512 steps,
                   run_scripts_examples/nemomocap-example.sh:10,17,30-33) through fit.run_fit, it/s per phase
  c3_bf16, c4   -- BASELINE configs[2] (40 x 300, bf16 operands) and configs[3] (256 x 1024), each with its own
                   roofline block; at N > 1 `c4` is the leg that has the work to scale
  scaling_model -- N = 1: what one rank of 2 / 4 / 8 computes per step (measured here) -> predicted speed-ups;
                   N > 1: per-rank compute and collective times of this run

N > 1: strong scaling -- the same 8 x 300 problem sharded by instance over the ranks (nemo_cvpr2023_amd/dist.py),
one process per GPU over RCCL.  The library default is ONE all-reduce per step (shared MLP gradient + loss scalars),
captured INSIDE the step's HIP graph; ``--shard-mode auto`` times the alternatives on the machine and keeps the fastest.

**A first multi-GPU run must not fail silently.**  Every rank process of an N > 1 run is a SUPERVISOR that never touches
the GPU: it starts the measuring worker as a child process (fresh interpreter, never a re-exec) under a wall-clock bound
and a no-progress watchdog, and the supervisors agree through the launcher's TCP store: if ANY rank's worker exits
non-zero, hangs or aborts (a failed RCCL-in-graph capture is the known risk), all workers are stopped and restarted with
``NEMO_GRAPH_COMM=0`` (graph, then eager collectives), then with ``NEMO_GRAPHS=0`` (no graphs at all).  Rank 0 prints the
surviving attempt's line with ``"fallback"`` / ``"graph_comm"`` / ``"attempts"`` filled in.  With no launcher environment
(``WORLD_SIZE`` unset) ``--gpus N`` starts the N supervisors itself through ``torch.distributed.run``.
Fewer visible GPUs than ``--gpus`` is an error (exit code 3), never a silent single-GPU number.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V0, T0 = 8, 300
# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks (fp32: v_mfma_f32_32x32x2_f32; bf16: 32x32x16) and HBM3E
MFMA_PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0, 'valu_f32': 157.3,      # (valu_f32: the fp32 vector peak = the fp32 MFMA peak)
                    'f16x3': 2500.0 / 3,      # fp32-equivalent products on the 16-bit pipe: three fp16 piece products each (mesh_blend f32_split)
                    'bf16x6': 2500.0 / 6,     # ... as six bf16 piece products (NEMO_MESH_PIECES=3, the round's first form)
                    'f16x4': 2500.0 / 4}      # ... as four fp16 piece products (the mesh kernel's skinnings, MODE 6)
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'traffic.json')
# what a supervisor tries, in order (environment of the worker processes)
ATTEMPTS = [('none', {}), ('NEMO_GRAPH_COMM=0', {'NEMO_GRAPH_COMM': '0'}),
            ('NEMO_GRAPHS=0', {'NEMO_GRAPH_COMM': '0', 'NEMO_GRAPHS': '0'})]


# Non-zero skinning weights per vertex of the synthetic SMPL-shaped model.  4 = the structure of the published SMPL model
# file (at most four joints influence a vertex), which the fused mesh kernel skins sparsely; 24 = a dense weight matrix (the
# `dense_skinning_weights` leg, and every round-1..3 record).  Both are the same operator (lbs.py:236-241) on different data.
SKIN_NNZ = 4


def step_flops(n, nv=6890, h=1000, din=105, skin_nnz=None, strict=False):
    """Algorithmic FLOPs of one published-configuration update step as THIS engine formulates it (DESIGN.md
    section 4): fused mesh term (2 pose blends, 2 skinnings, vertex->joint adjoint), blend-shape adjoint, MLP
    forward + activation-gradient + parameter-gradient GEMMs, VPoser encode/decode + KL backward, the
    pre-contracted joint GEMM and its adjoint, GMM prior.  Returns (total, per-part dict)."""
    nnz = SKIN_NNZ if skin_nnz is None else skin_nnz
    # (strict: the vertex->joint adjoint dA = W^T dT priced at the weight matrix's sparsity as well -- the kernel executes it as
    #  a dense 24-joint product on two MFMA joint tiles whatever the weights are)
    skin = 12 * (nnz if nnz <= 4 else 24)
    mesh = 2.0 * n * nv * (2 * 3 * 207 + 2 * skin + (skin if strict else 288))
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


def parse(argv=None):
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
    ap.add_argument('--no-extra-legs', action='store_true',
                    help='skip the legs beyond the headline: published_fit, c3_bf16, c4, the scaling model')
    ap.add_argument('--attempt-timeout', type=float, default=1500.0,
                    help='N > 1: wall-clock bound of one attempt of the rank workers before the supervisors fall back')
    ap.add_argument('--watchdog', type=float, default=300.0,
                    help='N > 1: a worker that makes no progress for this many seconds exits non-zero instead of hanging')
    ap.add_argument('--spawn-selftest', action='store_true',
                    help='test aid: the N ranks only rendezvous (gloo, no GPU) and report ranks_seen')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)      # a supervisor's child: the measuring process
    ap.add_argument('--attempt', default='main0', help=argparse.SUPPRESS)         # rendezvous prefix of this attempt
    ap.add_argument('--phase', choices=['main', 'legs'], default='main', help=argparse.SUPPRESS)
    ap.add_argument('--skin-nnz', type=int, default=SKIN_NNZ,
                    help='non-zero skinning weights per vertex of the synthetic body model (4 = as the published SMPL model; 24 = dense)')
    ap.add_argument('--cpu-child', default='', help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def spawn_ranks(opts):
    """Parent of a self-launched N-rank run.  Touches no GPU: ``device_count()`` does not initialise HIP on this image."""
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
    try:
        return subprocess.run(cmd, env=env, timeout=len(ATTEMPTS) * opts.attempt_timeout + 300).returncode
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: the rank supervisors did not finish inside their own bounds\n')
        return 4


def _kill_tree(proc):
    """Stop a worker and whatever it started (it runs in its own session / process group)."""
    if proc.poll() is not None:
        return
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            break
        try:
            proc.wait(timeout=10)
            break
        except subprocess.TimeoutExpired:
            continue


def rank_supervisor(opts):
    """One per rank of an N > 1 run.  Never touches the GPU, never initialises a process group: it runs the measuring
    worker as a child and agrees with the other ranks' supervisors through the launcher's TCP store on whether an
    attempt succeeded everywhere.  Two phases: `main` (the headline line; falls back through ATTEMPTS) and `legs` (the
    C4 leg, in the launch structure that survived; its failure costs the `c4` key, never the headline).  Returns the
    process exit code; rank 0 prints the JSON line."""
    import torch.distributed as dist
    from datetime import timedelta
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    try:
        store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), world_size=None, is_master=False,
                              timeout=timedelta(seconds=60))
        store = dist.PrefixStore('nemo_bench_supervisor', store)
    except Exception as ex:           # no launcher store to agree through: run unsupervised (one attempt, as before round 4)
        sys.stderr.write(f'bench.py[rank {rank}]: no launcher store ({ex!r}); running without the fallback supervisor\n')
        return worker_main(opts)
    me = os.path.abspath(__file__)
    argv = [a for a in sys.argv[1:] if a != '--worker']

    def run(tag, extra_env, phase):
        """One attempt of one phase on every rank.  -> (status, parsed JSON line of rank 0's worker or None)."""
        env = dict(os.environ, **extra_env)
        env['NEMO_BENCH_ATTEMPT'] = tag
        out = tempfile.NamedTemporaryFile('w+', prefix=f'nemo_bench_r{rank}_{tag}_', suffix='.out', delete=False)
        proc = subprocess.Popen([sys.executable, me, '--worker', '--attempt', tag, '--phase', phase] + argv, env=env,
                                stdout=out, start_new_session=True)
        t0, status = time.monotonic(), None
        fail_key, done_key = f'fail_{tag}', f'done_{tag}'
        while status is None:
            rc = proc.poll()
            if rc is not None:
                status = 'ok' if rc == 0 else f'rank {rank}: worker exit code {rc}'
            elif time.monotonic() - t0 > opts.attempt_timeout:
                status = f'rank {rank}: worker exceeded {opts.attempt_timeout:.0f} s'
            elif store.check([fail_key]):
                status = 'peer: ' + store.get(fail_key).decode()
            else:
                time.sleep(0.25)
        if status == 'ok':
            store.add(done_key, 1)
            # ... everywhere?  (a peer may still fail, e.g. while tearing its communicator down)
            while True:
                if store.check([fail_key]):
                    status = 'peer: ' + store.get(fail_key).decode()
                    break
                if store.add(done_key, 0) >= world:
                    break
                if time.monotonic() - t0 > opts.attempt_timeout + 120:
                    status = f'rank {rank}: peers did not finish attempt {tag}'
                    store.set(fail_key, status)
                    break
                time.sleep(0.25)
        elif not status.startswith('peer'):
            store.set(fail_key, status)
        _kill_tree(proc)
        line = None
        if status == 'ok' and rank == 0:
            out.seek(0)
            lines = [l for l in out.read().splitlines() if l.startswith('{')]
            line = json.loads(lines[-1]) if lines else None
            if line is None:
                status = 'rank 0: worker printed no JSON line'
        out.close()
        os.unlink(out.name)
        if status != 'ok':
            sys.stderr.write(f'bench.py[rank {rank}]: attempt {tag} failed -- {status}\n')
        return status, line

    log = []
    for k, (label, extra) in enumerate(ATTEMPTS):
        status, line = run(f'main{k}', extra, 'main')
        log.append({'attempt': k, 'env': label, 'status': status})
        if status != 'ok':
            continue
        legs = None
        if not opts.no_extra_legs and not opts.spawn_selftest:
            st2, legs = run(f'legs{k}', extra, 'legs')
            if st2 != 'ok':
                legs = {'c4': {'error': st2}}
        if rank == 0:
            line.update(fallback=label, graph_comm=(k == 0), attempts=log)
            line.update(legs or {})
            print(json.dumps(line), flush=True)
        return 0
    if rank == 0:
        print(json.dumps({'metric': 'NeMo fit iters/sec (instances x frames/step), Baseball-Pitch', 'value': None,
                          'unit': 'iters/s', 'n_gpus': world, 'error': 'every launch structure failed', 'attempts': log}),
              flush=True)
    return 6


class Watchdog:
    """A worker of an N > 1 run exits non-zero when it makes no progress for `seconds` (a hung collective or capture
    would otherwise sit there until the supervisor's bound): `beat()` at every phase boundary."""

    def __init__(self, seconds):
        self.seconds, self.last, self.what, self.limit = float(seconds), time.monotonic(), 'start', float(seconds)
        if seconds > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, what='', next_within=None):
        """A phase boundary has been reached; the NEXT one is due within `next_within` seconds (default: the global bound)."""
        self.last, self.what = time.monotonic(), what
        self.limit = self.seconds if next_within is None else min(self.seconds, float(next_within))

    def _run(self):
        while True:
            time.sleep(2.0)
            if time.monotonic() - self.last > self.limit:
                sys.stderr.write(f'bench.py watchdog: no progress for {self.limit:.0f} s after "{self.what}"; exiting 17\n')
                sys.stderr.flush()
                os._exit(17)


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


def init_group(dist, backend, rank, world, attempt):
    """Process group of attempt k: rendezvous keys under a prefix of their own (an earlier, failed attempt has left its
    keys in the launcher's store)."""
    from datetime import timedelta
    base = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), world_size=None, is_master=False,
                         timeout=timedelta(seconds=120))
    dist.init_process_group(backend, store=dist.PrefixStore(f'nemo_bench_attempt{attempt}', base), rank=rank, world_size=world)


# ----------------------------------------------------------------------------------------------- CPU leg
def cpu_child(spec):
    """``--cpu-child name,V,T,B,threads,warmup,steps,skin_nnz[,device]``: time the CPU oracle (child process, no GPU)."""
    name, V, T, B, threads, warm, steps, nnz, *dev = spec.split(',')
    V, T, B, threads, warm, steps, nnz = int(V), int(T), int(B), int(threads), int(warm), int(steps), int(nnz)
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
    o = OracleNemo(version, args, seqs, syn.make_smpl_assets(nv, seed=1, skin_nnz=nnz), syn.make_vposer_state(), syn.make_gmm())
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
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-child', f'{name},{V},{T},{B},{threads},{warm},{steps},{SKIN_NNZ}']
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


# ----------------------------------------------------------------------------------------------- measuring pieces
class Ctx:
    """What the legs of one worker share."""
    pass


def build_model(cx, V, T, dtype, batch_size=512, skin_nnz=None, weight_smooth=0.0, mesh_blend=None, locality=False):
    """(model, engine) for a V x T synthetic fit on this worker's device -- ShardedNemo over the ranks when sharded."""
    import torch
    from nemo_cvpr2023_amd import synthetic as syn
    args = syn.published_args(batch_size=batch_size, out_dir='')
    args.gemm_dtype = dtype
    # (profiling aid: every model of the run on the spatially structured synthetic body model -- tools/profile_r06.sh takes the mesh
    #  kernel's LDS / L1 counters on both body models this way; the line's `config.body_model` says which one ran)
    locality = locality or os.environ.get('NEMO_BENCH_LOCALITY') == '1'
    if mesh_blend:
        args.mesh_blend = mesh_blend            # 'f32': the mesh term's blend on the fp32 MFMA pipe (engine default: 'f32_split')
        args.mlp_gemm = mesh_blend              # ... and the MotionNet chain too: the `f32_mfma_blend` leg is the step on the fp32 pipe only
    if weight_smooth:
        args.weight_smooth = weight_smooth      # BASELINE configs[4]: the temporal-smoothness term in the loop
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets = dict(smpl_assets=syn.make_smpl_assets(6890, seed=1, skin_nnz=SKIN_NNZ if skin_nnz is None else skin_nnz, locality=locality),
                  vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    torch.manual_seed(0)
    if cx.sharded:
        from nemo_cvpr2023_amd.dist import ShardedNemo
        model = ShardedNemo(2, args, seqs, cx.device, rank=cx.rank, world=cx.world, seed=0, **assets)
        return model, model.model.engine, args
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    model = NemoV2(args, seqs, cx.device, **assets)
    return model, model.engine, args


def release(cx, model):
    import gc
    import torch
    if cx.sharded:
        model.close()                  # captured graphs hold RCCL launches: released before the communicator
    del model
    gc.collect()
    torch.cuda.empty_cache()


def timed(cx, fn, n):
    """EXACTLY n calls bracketed by barrier + synchronize on both sides; max over ranks."""
    import torch
    cx.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    cx.barrier()
    dt = time.perf_counter() - t0
    if cx.world > 1:
        tmax = torch.tensor([dt], device=cx.device, dtype=torch.float64)
        cx.dist.all_reduce(tmax, op=cx.dist.ReduceOp.MAX)
        dt = float(tmax)
    return dt, out


def snapshot(model):
    """Parameters, Adam moments and step counters of a model (restored around probes that take real update steps)."""
    mm = model.model if hasattr(model, 'model') else model
    e = mm.engine
    return (e.params.clone(), e.exp_avg.clone(), e.exp_avg_sq.clone(), [dict(o.steps) for o in mm.optimizers],
            [float(o.param_groups[0]['lr']) for o in mm.optimizers])


def restore(model, snap):
    mm = model.model if hasattr(model, 'model') else model
    e = mm.engine
    e.params.copy_(snap[0]); e.exp_avg.copy_(snap[1]); e.exp_avg_sq.copy_(snap[2])
    for o, st, lr in zip(mm.optimizers, snap[3], snap[4]):
        o.steps.update(st)
        o.param_groups[0]['lr'] = lr
    e.adam_table_invalidate()


def choose_shard_mode(cx, model, step, V, T, want):
    """--shard-mode auto: every candidate runs the SAME 4 + 10 update steps from the SAME state (snapshot / restore), so
    their last total_loss must agree -- a candidate with broken numerics cannot win on time alone."""
    modes = (['single', 'split'] + (['buckets'] if V * T // max(cx.world, 1) >= 4096 else [])) if want == 'auto' else [want]
    snap = snapshot(model)
    ms, last = {}, {}
    for mode in modes:
        restore(model, snap)
        model.set_shard_mode(mode)
        for _ in range(4):
            step()
        dt, out = timed(cx, step, 10)
        ms[mode], last[mode] = round(1e3 * dt / 10, 4), float(out[0]['total_loss'])
        cx.wd.beat(f'shard mode {mode}')
    ref = last[modes[0]]
    agree = {m_: abs(v - ref) <= 2e-3 * max(abs(ref), 1e-9) for m_, v in last.items()}
    ok = [m_ for m_ in modes if agree[m_]]
    best = min(ok, key=ms.get)       # max-over-ranks timings, identical losses on every rank: the same choice everywhere
    restore(model, snap)
    model.set_shard_mode(best)
    return best, ms, {m_: {'total_loss_after_14_steps': last[m_], 'agrees_with_' + modes[0]: agree[m_]} for m_ in modes}


def roofline_block(cx, engine, timers, n_inst, ms_per_step, V, T, dtype, skin_nnz=None, counters_apply=True):
    """The dominant tagged kernel of `timers` against the peak of its own mix of matrix pipes, + the whole step."""
    best = None
    for tag, evs in timers.items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        tot = sum(ms)
        if best is None or tot > best[0]:       # mean launch time and mean FLOPs per launch (chunked batches differ)
            best = (tot, tag, sum(ms) / len(ms), sum(e_[2] for e_ in evs) / len(evs), len(ms))
    if best is None:
        return None
    _, tag, mean_ms, flops, n = best
    # Price every part of the kernel against the pipe it runs on: with bf16 the pose blends and the vertex->joint adjoint
    # of the fused mesh kernel are bf16 work, its skinning / L1 stay on the fp32 pipe.  The peak quoted is the rate at
    # which the kernel's own mix of work would run with both pipes at their peaks (harmonic mix; = the fp32 peak for
    # the fp32 build).
    f_step, parts = step_flops(V * T // cx.world if cx.world > 1 else V * T, skin_nnz=skin_nnz)
    pipes = engine.kernel_flops_by_pipe(tag, flops)
    kpeak = flops / sum(f / MFMA_PEAK_TFLOPS[d] for d, f in pipes.items())
    achieved = flops / (mean_ms * 1e-3) / 1e12
    traffic = {}
    # (counters were collected on the default body model and loss terms: legs on another body model / with further terms quote none)
    if os.path.exists(TRAFFIC_FILE) and skin_nnz in (None, SKIN_NNZ) and counters_apply and os.environ.get('NEMO_BENCH_LOCALITY') != '1':
        traffic = json.load(open(TRAFFIC_FILE)).get(f'{V}x{T}x{cx.world}x{dtype}', {})
    # counters are only quoted for the kernel instantiation they were taken on (the file records it, with the commit)
    variant = engine.mesh_kernel_variant() if tag == 'mesh_v2v_fused' else None
    stale = variant is not None and traffic.get('kernel_variants', {}).get(tag) not in (None, variant)
    ktr = None if stale else traffic.get('kernels', {}).get(tag)
    busy = None if stale else traffic.get('mfma_busy', {}).get(tag)
    # the same kernel with the adjoint dA = W^T dT priced at the weights' real sparsity (`frac_strict`)
    f_strict = flops
    if tag == 'mesh_v2v_fused':
        f_strict = flops * engine.mesh_macs(strict=True) / engine.mesh_macs()
    mesh_b16 = engine.kernel_flops_by_pipe('mesh_v2v_fused', parts['mesh']).get('bf16', 0.0)
    on_bf16 = (mesh_b16 + parts['blend_adjoint'] + parts['mlp'] + parts['vposer']) if dtype == 'bf16' else 0.0
    _mp = engine.kernel_flops_by_pipe('mesh_v2v_fused', parts['mesh'])
    on_b16x6, on_f16x3, on_f16x4 = _mp.get('bf16x6', 0.0), _mp.get('f16x3', 0.0), _mp.get('f16x4', 0.0)
    if dtype != 'bf16' and getattr(engine, 'split_adj', False) and engine.ctx.split_ok:
        on_f16x3 += parts['blend_adjoint']              # (the blend-shape adjoint in split precision too)
    if dtype != 'bf16' and getattr(engine, 'mlp_split', False) and any('Xx' in w_ for w_ in engine.ws.values()):
        # (the MotionNet chain on nemo_gemm_xp: three fp16 / six bf16 piece products per algorithmic product)
        if engine.xp_fmt == 2:
            on_f16x3 += parts['mlp']
        else:
            on_b16x6 += parts['mlp']
    step_peak = f_step / (on_bf16 / MFMA_PEAK_TFLOPS['bf16'] + on_b16x6 / MFMA_PEAK_TFLOPS['bf16x6'] + on_f16x3 / MFMA_PEAK_TFLOPS['f16x3'] +
                          on_f16x4 / MFMA_PEAK_TFLOPS['f16x4'] + (f_step - on_bf16 - on_b16x6 - on_f16x3 - on_f16x4) / MFMA_PEAK_TFLOPS['f32'])
    roof = {'kernel': tag, 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(kpeak, 1),
            'unit': 'TFLOP/s', 'frac': round(achieved / kpeak, 4),
            'frac_strict': round(achieved * (f_strict / flops) / kpeak, 4),
            'frac_strict_note': 'the vertex->joint adjoint dA = W^T dT priced at the skinning weights\' real sparsity (4 of 24 joints) '
                                'instead of the dense 24-joint product the kernel executes; = frac for a dense body model',
            'mfma_busy': busy, 'mfma_busy_source': traffic.get('mfma_busy_source') if busy is not None else None,
            'pipes': {d: round(f / 1e9, 2) for d, f in pipes.items()},
            'peak_note': 'fp32 MFMA peak' if set(pipes) <= {'f32', 'valu_f32'} else
                         'harmonic mix of the per-pipe peaks over this kernel\'s algorithmic GFLOP per pipe (`pipes`): bf16 MFMA 2500, '
                         'fp32 MFMA / fp32 VALU 157.3, f16x3 = fp32-equivalent products as three fp16 piece products = 2500 / 3 '
                         '(mesh_blend f32_split, mlp_gemm f32_split), f16x4 = four fp16 piece products = 2500 / 4 (the mesh kernel\'s skinnings), '
                         'bf16x6 = six bf16 piece products = 2500 / 6',
            'traffic': ktr, 'traffic_source': traffic.get('source') if ktr else None,
            'traffic_commit': traffic.get('commit') if ktr else None, 'traffic_kernel': variant if ktr else None,
            'traffic_dropped': 'counters in profiles/traffic.json were taken on ' + str(traffic.get('kernel_variants', {}).get(tag)) +
                               ', this run launches ' + str(variant) if stale else None,
            'launches': n, 'mean_launch_ms': round(mean_ms, 4), 'flops_per_launch': flops,
            'timed_in': f'{n_inst} instrumented (un-captured) steps after the timed region',
            'per_kernel_ms_per_step': {t: round(sum(a.elapsed_time(b) for a, b, _ in e_) / n_inst, 4)
                                       for t, e_ in timers.items()},
            # the whole step against both roofs: algorithmic FLOPs of one rank's step / step time against the peak of
            # its own mix of pipes (fp32 build: the fp32 MFMA peak), and counter-measured HBM-side bytes per step
            # (separate --pmc passes, profiles/) / step time
            'step': {'flops': f_step, 'achieved': round(f_step / (ms_per_step * 1e-3) / 1e12, 2),
                     'peak': round(step_peak, 1), 'unit': 'TFLOP/s',
                     'frac': round(f_step / (ms_per_step * 1e-3) / 1e12 / step_peak, 4),
                     'gflop_on_bf16_pipe': round(on_bf16 / 1e9, 2),
                     'flops_parts': {k: round(v / 1e9, 2) for k, v in parts.items()}},
            'hbm': None}
    # the step's counter bytes are only quoted for the build they were counted on: mesh kernel variant + MotionNet-chain arithmetic
    # (profiles/traffic.json records both as `step_variant`)
    step_variant = engine.mesh_kernel_variant() + '|mlp=' + (('f32_split' if engine.xp_fmt == 2 else 'f32_split3') if (
        getattr(engine, 'mlp_split', False) and any('Xx' in w_ for w_ in engine.ws.values())) else ('bf16' if engine.bf16 else 'f32'))
    roof['step_variant'] = step_variant
    if traffic.get('step_bytes') and traffic.get('step_variant') not in (None, step_variant):
        roof['hbm_dropped'] = f"step bytes in profiles/traffic.json were counted on {traffic.get('step_variant')}, this run is {step_variant}"
    elif traffic.get('step_bytes'):
        gbs = traffic['step_bytes'] / (ms_per_step * 1e-3) / 1e9
        roof['hbm'] = {'bytes_per_step': traffic['step_bytes'], 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                       'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'source': traffic.get('source')}
    roof['hbm_frac'] = roof['hbm']['frac'] if roof['hbm'] else None
    return roof


def instrumented(cx, engine, step, n_inst):
    """The production step replays a captured HIP graph (events cannot be recorded inside one), so the same steps are run
    once more, un-captured, with HIP events around the tagged kernels on their launch streams.  Not part of `value`."""
    engine.timers = {}
    for _ in range(n_inst):
        step()
    cx.barrier()
    timers, engine.timers = engine.timers, None
    return timers


def leg(cx, name, V, T, dtype, steps, warm, shard_mode='single', skin_nnz=None, weight_smooth=0.0, mesh_blend=None, locality=False):
    """A further BASELINE configuration as an extra key of the line: full-batch update steps of a V x T fit, timed like
    the headline (barrier + synchronize, max over ranks), with its own roofline block."""
    import torch
    try:
        model, engine, _ = build_model(cx, V, T, dtype, skin_nnz=skin_nnz, weight_smooth=weight_smooth, mesh_blend=mesh_blend, locality=locality)
        if cx.sharded:
            model.set_shard_mode(shard_mode)

        def step():
            return model.step(None, None, update=True, full_batch=True)
        for _ in range(2):               # set-up: workspace + graph capture
            step()
        for _ in range(warm):
            step()
        # two timed regions, the faster one reported (both listed): a leg that starts right behind a heavy one (C4: 115 ms
        # steps) has been seen 15 - 30 % slow for its first tens of milliseconds
        regions = [timed(cx, step, steps) for _ in range(2)]
        dt, out = min(regions, key=lambda r_: r_[0])
        ms = 1e3 * dt / steps
        timers = instrumented(cx, engine, step, 2)
        res = {'value': round(steps / dt, 3), 'unit': 'iters/s', 'ms_per_step': round(ms, 3), 'steps': steps, 'warmup': warm,
               'regions_ms_per_step': [round(1e3 * r_[0] / steps, 3) for r_ in regions],
               'dtype': dtype, 'samples_per_s': round(V * T * steps / dt, 1),
               'workload': f'{V} instances x {T} frames full batch (N={V * T}), published hyper-parameters, all loss terms'
                           + (f' + temporal smoothness of the output joints (weight {weight_smooth:g})' if weight_smooth else ''),
               'final_total_loss': float(out[0]['total_loss']),
               **({'final_smooth_loss': float(out[0]['smooth_loss'])} if weight_smooth else {}),
               'skinning': 'sparse' if engine.ctx.skin_sparse else 'dense', 'skin_nnz': engine.ctx.skin_nnz,
               'body_model': 'synthetic, spatially structured (synthetic.make_smpl_assets(locality=True))' if locality else 'synthetic, random permutation (default)',
               'mesh_kernel': engine.mesh_kernel_variant(),
               'roofline': roofline_block(cx, engine, timers, 2, ms, V, T, dtype, skin_nnz=skin_nnz,
                                          counters_apply=not (locality or weight_smooth))}
        if cx.sharded:
            res['shard_mode'] = model.shard_mode
            res.update(shard_probe(cx, model, step, steps))
        release(cx, model)
        cx.wd.beat(f'leg {name}')
        return res
    except Exception as ex:          # a leg must never take the headline down
        import traceback
        traceback.print_exc()
        if cx.world > 1:
            raise                    # (ranks must stay in step: let the supervisor fall back)
        return {'error': repr(ex)[:300]}


def shard_probe(cx, model, step, steps):
    """What one rank computes per step without any collective, and what the collective costs alone -- so that a multi-GPU
    number can be read as  step = compute + (un-hidden part of the) collective.  The compute-only steps would let the
    replicas drift apart (Adam on un-reduced gradients): the model's state is restored afterwards."""
    import torch
    dist = cx.dist
    n_probe = max(5, min(50, steps))
    snap = snapshot(model)
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
        restore(model, snap)
    a_, b_ = model._span
    buf = torch.zeros(b_ - a_, device=cx.device)
    for _ in range(5):
        dist.all_reduce(buf)
    cx.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    coll_ms = 1e3 * (time.perf_counter() - t0) / 20
    per_rank = [None] * cx.world
    dist.all_gather_object(per_rank, {'rank': cx.rank, 'compute_ms': round(compute_ms, 4), 'collective_ms': round(coll_ms, 4),
                                      'instances': model.plan.v_local, 'device': torch.cuda.current_device()})
    worst = max(p_['compute_ms'] for p_ in per_rank)
    coll = max(p_['collective_ms'] for p_ in per_rank)
    return {'per_rank': per_rank, 'allreduce_bytes': int((b_ - a_) * 4),
            'scaling_model': {'shard_step_ms': worst, 'allreduce_ms': coll,
                              'predicted_ms_per_step_no_overlap': round(worst + coll, 4),
                              'note': 'compute_ms = this rank\'s step with every collective stubbed out; collective_ms = '
                                      'the all-reduce of the shared-gradient slice alone (back-to-back, synchronised); '
                                      'the measured ms_per_step lies between max(compute) and compute + collective'}}


def scaling_model_single_gpu(cx, headline_ms, c4_ms):
    """N = 1: what ONE rank of a 2 / 4 / 8-GPU run computes per step, measured here (the k-instance problem on this
    GPU), and the speed-up that predicts once a 9 MB gradient all-reduce over xGMI is added un-hidden.  The headline
    (8 x 300: a 1.5 ms step) is latency-bound -- one rank's share is a third of the step, not an eighth --, C4
    (256 x 1024) is where the work to scale is."""
    out = {'assumed_allreduce_ms': [0.1, 0.3],
           'note': 'per_rank_compute_ms[W] = measured step time of one rank\'s share (V / W instances) on this GPU; '
                   'predicted_speedup[W] = single-GPU step / (per-rank compute + an un-hidden 9 MB all-reduce at the two assumed '
                   'costs); xGMI all-reduce times are NOT measured here (one GPU)'}
    for name, V, T, base_ms, steps in (('headline_8x300', V0, T0, headline_ms, 30), ('c4_256x1024', 256, 1024, c4_ms, 3)):
        if base_ms is None:
            continue
        per_rank, pred, regions = {}, {}, {}
        for W in (2, 4, 8):
            try:
                model, engine, _ = build_model(cx, V // W, T, 'f32')

                def step():
                    return model.step(None, None, update=True, full_batch=True)
                for _ in range(5):
                    step()
                # two timed regions, the minimum counts (BENCH_r04: one region after 3 set-up steps caught a 73 ms stall behind
                # the previous model's release and reported 2.9 ms for a 0.47 ms step); every step of them must be a graph replay
                model.launch_stats['replayed'] = model.launch_stats['other'] = 0
                # (round 6: a region that disagrees with the other -- the stall lands in either, two of ten default runs had it in the
                #  second -- is repeated, up to four regions, until the two fastest agree within 10 %)
                regs = []
                while len(regs) < 4:
                    regs.append(1e3 * timed(cx, step, steps)[0] / steps)
                    two = sorted(regs)[:2]
                    if len(regs) >= 2 and two[1] <= 1.1 * two[0]:
                        break
                two = sorted(regs)[:2]
                ls = model.launch_stats
                frac = ls['replayed'] / max(1, ls['replayed'] + ls['other'])
                per_rank[W] = round(min(regs), 4)
                regions[W] = {'ms_per_step': [round(r_, 4) for r_ in regs], 'graph_replay_fraction': round(frac, 3),
                              'suspect': bool(two[1] > 1.1 * two[0] or frac < 1.0)}
                pred[W] = [round(base_ms / (per_rank[W] + a_), 2) for a_ in out['assumed_allreduce_ms']]
                release(cx, model)
            except Exception as ex:
                per_rank[W] = {'error': repr(ex)[:200]}
            cx.wd.beat(f'scaling model {name} W={W}')
        out[name] = {'single_gpu_ms_per_step': round(base_ms, 4), 'per_rank_compute_ms': per_rank, 'predicted_speedup': pred,
                     'regions': regions, 'suspect': any(r_.get('suspect') for r_ in regions.values())}
    return out


def published_fit(cx):
    """The published schedule end to end (nemomocap-example.sh:10,17,30-33): eval at init, 300 warm-up iterations, 1000
    camera-fit iterations, 2000 minibatch-512 steps with the script's evaluation cadence -- through fit.run_fit."""
    import torch
    from nemo_cvpr2023_amd import fit
    try:
        walls, runs = [], []
        for rep in range(2):                  # the schedule twice, each on a freshly built model from the same seeds; the faster
            model, engine, args = build_model(cx, V0, T0, 'f32')       # run is reported, both are listed
            args.warmup_step, args.opt_cam_step, args.n_steps = 300, 1000, 2000
            torch.manual_seed(0)
            # set-up outside the clock (like the capture steps of the headline): workspaces + graphs of the three phases
            model.warmup(3); model.opt_cam(3)
            for _ in range(3):
                model.step(*model.draw_batch())
            torch.manual_seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r_ = fit.run_fit(model, args, out_dir=None, evaluate=None)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            runs.append(r_)
            release(cx, model)
            cx.wd.beat(f'published_fit run {rep}')
        res = runs[walls.index(min(walls))]
        wall = min(walls)
        ph = res['phase_seconds']
        out = {'wall_seconds': round(wall, 3), 'wall_seconds_runs': [round(w_, 3) for w_ in walls],
               'suspect': bool(max(walls) > 1.3 * min(walls)),
               'final_cam_loss_runs': [r_['cam_losses'][-1] for r_ in runs],
               'final_total_loss_runs': [r_['losses']['total_loss'][-1] for r_ in runs],
               'schedule': {'warmup': 300, 'opt_cam': 1000, 'steps': 2000, 'batch': 512},
               'phases': {k: {'seconds': round(v, 4), 'iters': n, 'iters_per_s': round(n / v, 1), 'ms_per_iter': round(1e3 * v / n, 4)}
                          for (k, v), n in zip(ph.items(), (300, 1000, 2000))},
               'final_total_loss': res['losses']['total_loss'][-1], 'final_warmup_loss': res['warmup_losses'][-1],
               'final_cam_loss': res['cam_losses'][-1],
               'note': 'fit.run_fit on 8 x 300 synthetic sequences: host-drawn batches (CPU RNG, the script\'s order), evaluation '
                       'steps at step 0 and every 500th, no checkpoints / metrics written; warm-up and camera fit run as '
                       'captured iterations whose losses are read once per phase'}
        return out
    except Exception as ex:
        import traceback
        traceback.print_exc()
        return {'error': repr(ex)[:300]}


# ----------------------------------------------------------------------------------------------- the measuring process
def _capture_fault(sharded, comm_inside):
    """Fault injection for the tests of the multi-GPU fallback (NEMO_TEST_FAIL_CAPTURE = raise | exit | hang): the first capture
    of a sharded step on rank 1 of a supervisor's first attempt raises, kills the process or hangs.  Installed into the
    product's capture seam (neural_motion_model._capture_hook) by worker_main, only when the variable is set."""
    fault = os.environ.get('NEMO_TEST_FAIL_CAPTURE')
    if not fault or not sharded:
        return
    if fault == 'raise' and comm_inside:
        raise RuntimeError('NEMO_TEST_FAIL_CAPTURE=raise')
    if os.environ.get('NEMO_BENCH_ATTEMPT') == 'main0' and os.environ.get('RANK') == '1':
        if fault == 'exit':
            os._exit(23)
        if fault == 'hang':
            time.sleep(3600)


def worker_main(opts):
    if os.environ.get('NEMO_TEST_FAIL_CAPTURE'):
        from nemo_cvpr2023_amd import neural_motion_model as _nmm
        _nmm._capture_hook = _capture_fault
        print(f"bench.py: FAULT INJECTION ARMED (NEMO_TEST_FAIL_CAPTURE={os.environ['NEMO_TEST_FAIL_CAPTURE']})", file=sys.stderr, flush=True)
    V, T = opts.instances, opts.frames
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != opts.gpus:
        raise SystemExit(f'bench.py: --gpus {opts.gpus} but WORLD_SIZE={world}')
    attempt = opts.attempt

    import torch
    import torch.distributed as dist
    cx = Ctx()
    cx.wd = Watchdog(opts.watchdog if world > 1 else 0)

    if opts.spawn_selftest:                                   # rendezvous only (CPU test of the launcher)
        fault = os.environ.get('NEMO_TEST_FAIL_CAPTURE', '')
        if opts.worker:
            init_group(dist, 'gloo', rank, world, attempt)
        else:
            dist.init_process_group('gloo')
        ranks = check_world(opts, dist, torch, rank, world, local_rank)
        if fault and attempt == 'main0' and rank == 1:        # fault injection: this rank's first capture dies / hangs
            if fault == 'hang':
                time.sleep(3600)
            os._exit(23)
        dist.barrier()
        if rank == 0:
            print(json.dumps({'selftest': True, 'n_gpus': world, 'ranks_seen': ranks}), flush=True)
        dist.destroy_process_group()
        return 0

    from nemo_cvpr2023_amd import synthetic as syn            # noqa: F401  (import errors before any rendezvous)
    have = torch.cuda.device_count()
    backend = os.environ.get('NEMO_DIST_BACKEND', 'nccl')
    if have < (world if backend == 'nccl' else 1):
        raise SystemExit(f'bench.py: {world} ranks but {have} GPU(s) visible')
    cx.device = f'cuda:{local_rank % have}'
    torch.cuda.set_device(cx.device)
    cx.rank, cx.world, cx.dist = rank, world, dist
    ranks_seen = [0]
    # diagnostic: NEMO_BENCH_SHARD_OF_ONE=1 runs the SHARDED code path (ShardedNemo, the all-reduce, Adam after it, every
    # shard mode) in a process group of one rank -- what the sharded host / launch structure costs by itself on one GPU
    cx.sharded = world > 1
    if world == 1 and os.environ.get('NEMO_BENCH_SHARD_OF_ONE') == '1':
        cx.sharded = True
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        dist.init_process_group(backend, rank=0, world_size=1)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL on ROCm.  (NEMO_DIST_BACKEND=gloo: test aid for boxes where several ranks must share one GPU,
        # which RCCL refuses.)
        if opts.worker:
            init_group(dist, backend, rank, world, attempt)
        else:
            dist.init_process_group(backend)
        ranks_seen = check_world(opts, dist, torch, rank, world, local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    cx.barrier = barrier

    if opts.phase == 'legs':         # (a supervisor's second child: the C4 leg alone, as {'c4': {...}})
        res = {'c4': leg(cx, 'c4', 256, 1024, 'f32', 3, 2)}
        if rank == 0:
            print(json.dumps(res), flush=True)
        dist.destroy_process_group()
        return 0

    model, engine, args = build_model(cx, V, T, opts.dtype)
    # (the first sharded steps hold the one thing no box available to this build has ever run: an RCCL collective between
    #  two GPUs, captured into a HIP graph -- if they hang, give up on this attempt after two minutes, not seven)
    cx.wd.beat('model built', next_within=120 if cx.sharded else None)

    def step():
        return model.step(None, None, update=True, full_batch=True)

    # set-up: the first calls of a (batch size, mode) variant allocate its workspace and capture its HIP graph
    # (like a compile step); they are not part of the warm-up / timed protocol below
    shard_modes = mode_check = None
    if cx.sharded:
        # auto: one collective per step against the early loss all-reduce against three gradient buckets behind the backward.
        # With RCCL every mode is ONE captured launch per step (collectives inside the graph); the three-bucket layout is
        # only tried for long per-rank steps: in a group of one it needs > 150 us of hidden collective per step to catch up
        # with `split` (0.67 against 0.52 ms at one instance), more than the whole 9 MB all-reduce takes
        for _ in range(2):
            step()
        cx.wd.beat('first sharded steps')
        _, shard_modes, mode_check = choose_shard_mode(cx, model, step, V, T, opts.shard_mode)
    for _ in range(3):
        step()
    for _ in range(opts.warmup):
        step()
    cx.wd.beat('warm-up')
    # EXACTLY --steps steps per timed region (barrier + synchronize on both sides, max over ranks); --repeat regions, the
    # median one is reported (all of them are listed in `repeat_ms_per_step`)
    regions = [timed(cx, step, opts.steps) for _ in range(max(1, opts.repeat))]
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, (ld, _) = regions[order[len(order) // 2]]
    repeat_ms = [round(1e3 * r[0] / opts.steps, 4) for r in regions]
    ms_per_step = 1e3 * elapsed / opts.steps
    iters_per_s = opts.steps / elapsed
    cx.wd.beat('timed regions')

    # ---- the published run's mode: random minibatches of 512 samples drawn like the script does (CPU RNG, views then frames)
    mini = None
    if opts.minibatch_steps > 0 and V * T >= 512:
        gen = torch.Generator().manual_seed(1234)
        draws = [(torch.randint(0, V, (512,), generator=gen), torch.randint(0, T, (512,), generator=gen))
                 for _ in range(opts.minibatch_steps + 12)]
        mm = model.model if cx.sharded else model
        for vi_, fi_ in draws[:12]:                     # set-up: workspaces + graph captures of the launch sizes
            model.step(vi_, fi_)
        stats0 = dict(mm.launch_stats)
        it = iter(draws[12:])
        dt, _ = timed(cx, lambda: model.step(*next(it)), opts.minibatch_steps)
        rep = mm.launch_stats['replayed'] - stats0['replayed']
        oth = mm.launch_stats['other'] - stats0['other']
        mini = {'value': round(opts.minibatch_steps / dt, 2), 'unit': 'iters/s', 'ms_per_step': round(1e3 * dt / opts.minibatch_steps, 4),
                'batch': 512, 'steps': opts.minibatch_steps, 'graph_replay_fraction': round(rep / max(rep + oth, 1), 3),
                'note': 'random (view, frame) minibatches of 512 out of %d x %d, drawn on the host like '
                        'scripts/learned_multi_view_recon_nn.py:291-296; rank 0\'s launches' % (V, T)}
        cx.wd.beat('minibatch leg')

    n_inst = max(3, min(10, opts.steps))
    timers = instrumented(cx, engine, step, n_inst)
    roof = roofline_block(cx, engine, timers, n_inst, ms_per_step, V, T, opts.dtype)
    cx.wd.beat('roofline leg')

    # ---- sharded runs: per-rank compute / collective (LAST on this model: the probe takes un-reduced update steps)
    shard_info = shard_probe(cx, model, step, opts.steps) if cx.sharded else None
    shard_mode = model.shard_mode if cx.sharded else None
    mesh_split, mesh_variant = bool(engine.mesh_split), engine.mesh_kernel_variant()
    engine_blend = engine.mesh_blend_in_effect()
    engine_mlp = 'bf16' if engine.bf16 else (('f32_split' if engine.xp_fmt == 2 else 'f32_split3') if engine.mlp_split else 'f32')
    del engine
    release(cx, model)
    cx.wd.beat('headline done')

    # ---- the other BASELINE configurations and the end-to-end schedule, as extra keys of the same line
    extra = {}
    if not opts.no_extra_legs and (V, T, opts.dtype) == (V0, T0, 'f32'):
        if world == 1 and not cx.sharded:
            extra['published_fit'] = published_fit(cx)
            extra['c3_bf16'] = leg(cx, 'c3_bf16', 40, 300, 'bf16', 10, 3)
        if world == 1:               # (N > 1: the supervisor runs this leg as a run of its own, see rank_supervisor)
            extra['c4'] = leg(cx, 'c4', 256, 1024, 'f32', 3, 2, shard_mode=shard_mode or 'single')
        if world == 1 and not cx.sharded:
            extra['scaling_model'] = scaling_model_single_gpu(cx, ms_per_step, extra['c4'].get('ms_per_step'))
            # BASELINE configs[4] on one GPU: the headline sizes with the temporal-smoothness term in the loop as well
            extra['c5_smooth'] = leg(cx, 'c5_smooth', V0, T0, 'f32', 20, 3, weight_smooth=1e5)
            if SKIN_NNZ <= 4:
                # the headline workload with a DENSE skinning-weight matrix (24 non-zero weights per vertex: the synthetic
                # model of rounds 1-3; the mesh kernel then runs the 24-joint product on the MFMA pipe) -- continuity
                # with the earlier records, and what a body model without SMPL's sparsity would cost
                extra['dense_skinning_weights'] = leg(cx, 'dense_skinning_weights', V0, T0, 'f32', 20, 3, skin_nnz=24)
            # the headline workload with the mesh term's pose blend on the fp32 MFMA pipe (the arithmetic of rounds 1 - 4); the
            # headline itself runs the engine default, the fp32-equivalent three-piece bf16 blend (`f32_split` below)
            extra['f32_mfma_blend'] = leg(cx, 'f32_mfma_blend', V0, T0, 'f32', 20, 3, mesh_blend='f32')
            # the headline workload on a body model with SMPL's spatial structure (vertices ordered by body part, 1 - 2 dominant
            # skinning weights, sparse local joint regressors): every other number of this line is on the default synthetic model,
            # whose vertices are a random permutation -- pessimal for the sparse skinning's LDS reads and the mesh kernel's L1
            extra['locality_body_model'] = leg(cx, 'locality_body_model', V0, T0, 'f32', 20, 3, locality=True)

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
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-child', f'c2,{V},{T},-1,0,2,{n_t},{SKIN_NNZ},cuda']
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
                                   'NemoV2 published hyper-parameters, all loss terms, 6890-vertex SMPL-shaped model with '
                                   + (f'{SKIN_NNZ} non-zero skinning weights per vertex (the published SMPL model\'s sparsity)'
                                      if SKIN_NNZ <= 4 else 'a dense skinning-weight matrix'),
                       'instances': V, 'frames': T, 'samples_per_step': V * T, 'h_dim': args.h_dim, 'skin_nnz': SKIN_NNZ,
                       'mesh_blend': engine_blend, 'mlp_gemm': engine_mlp,
                       'body_model': ('synthetic, spatially structured (NEMO_BENCH_LOCALITY=1)' if os.environ.get('NEMO_BENCH_LOCALITY') == '1'
                                      else 'synthetic, random vertex permutation (default; the `locality_body_model` key is the same step on the structured one)'),
                       'mesh_kernel': mesh_variant,
                       'parallelism': f'instance-shard x{world}' if world > 1 else
                       ('sharded code path in a process group of ONE rank (diagnostic)' if cx.sharded else 'single GPU')},
            'ranks_seen': ranks_seen,
            'repeat': len(repeat_ms), 'repeat_ms_per_step': repeat_ms,
            'minibatch512': mini,
            'final_total_loss': float(ld['total_loss']),
            'roofline': roof, 'cpu_baseline': cpu, 'torch_gpu_baseline': tgpu,
        }
        out.update(extra)
        if opts.dtype == 'f32' and mesh_split:
            fm = extra.get('f32_mfma_blend') or {}
            out['f32_split'] = {
                'what': 'the headline runs the fused mesh term with its pose blend (3 x 207 multiply-adds per vertex, body and sample) and its '
                        'vertex->joint adjoint on the 16-bit matrix cores in fp32-equivalent arithmetic: both operands as two fp16 pieces of '
                        's x (11 + 11 significant bits + the remainder\'s sign = the fp32 value to one ulp), the three leading piece products '
                        'exact in fp32, fp32 accumulation (nemo_v2v_fused_split / _splitmem, csrc/smpl.hip MODE 5), and the blend-shape adjoint GEMM '
                        'behind it the same way (nemo_gemm_f16x2mem_adj, from 256 samples on); round 6: the nn.Linear products of MotionNet '
                        '(forward, dX, dW) likewise (nemo_gemm_xp fmt 2, from 1600 rows on; the range guard of the fp16 pieces is device-side: '
                        'scale records, see include/nemo_hip.h); everything else is the fp32 path.  '
                        '`f32_mfma_blend` is the same step with ALL these products on the fp32 MFMA pipe (mesh_blend = mlp_gemm = f32)',
                'criteria': {'a_error_vs_float64': 'tests/test_gpu_ops.py::test_v2v_fused_split_is_fp32_equivalent: error of loss, d vp and dA '
                                                   'against a float64 evaluation of the same fp32 inputs <= 1.5 x the fp32-MFMA kernel\'s '
                                                   '(measured: dA rms 1.567e-7 against 1.578e-7, blend shapes x 100); the arithmetic itself: '
                                                   'tests/test_split_precision.py',
                             'a2_mlp_error_vs_float64': 'tests/test_gpu_xp.py: nemo_gemm_xp error against float64 <= 1.5 x nemo_gemm_f32\'s in every '
                                                        'role of the chain, operand magnitudes 1e-30 ... 1e30; the chain inside the step against the '
                                                        'fp32 chain: losses to 2e-6, gradients to 2e-5 of their scale',
                             'b_parity_gates': 'every 1e-4 parity gate of tests/ runs on this default, unchanged',
                             'c_launch_time': 'profiles/r05_f32_split.md: 316 against 519 us per 8 x 300 launch (-39 %); this run: '
                                              'mesh_launch_ms against mesh_launch_ms_f32_mfma_blend below'},
                'ms_per_step': round(ms_per_step, 3), 'ms_per_step_f32_mfma_blend': fm.get('ms_per_step'),
                'mesh_launch_ms': (roof or {}).get('mean_launch_ms'),
                'mesh_launch_ms_f32_mfma_blend': (fm.get('roofline') or {}).get('mean_launch_ms')}
        if cx.sharded:
            out['backend'] = dist.get_backend()
            out['shard_modes_ms'] = shard_modes
            out['shard_mode_check'] = mode_check
            out['shard_mode'] = shard_mode
            out['collectives_per_step'] = {'single': 1, 'split': 2, 'buckets': 3}[shard_mode]
            out['graph_comm'] = os.environ.get('NEMO_GRAPH_COMM', '1') != '0' and os.environ.get('NEMO_GRAPHS', '1') != '0'
            out.update(shard_info)
            if world > 1:
                out['scaling_note'] = ('the headline (8 x 300, a 1.5 ms single-GPU step) is latency-bound: one rank\'s share of an 8-way '
                                       'split still takes ~0.48 ms before any wire time (profiles/r03_shard_budget.md), i.e. <= ~3x; '
                                       'the `c4` key (256 x 1024) is the configuration with the work to scale')
        print(json.dumps(out), flush=True)
    if cx.sharded:
        dist.destroy_process_group()
    return 0


def main():
    global SKIN_NNZ
    opts = parse()
    SKIN_NNZ = opts.skin_nnz
    if opts.cpu_child:
        return cpu_child(opts.cpu_child)
    if opts.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if opts.worker:
        return worker_main(opts)
    if opts.gpus > 1:
        if 'WORLD_SIZE' not in os.environ:
            return spawn_ranks(opts)
        if int(os.environ['WORLD_SIZE']) != opts.gpus:
            raise SystemExit(f'bench.py: --gpus {opts.gpus} but WORLD_SIZE={os.environ["WORLD_SIZE"]}')
        return rank_supervisor(opts)
    return worker_main(opts)


if __name__ == '__main__':
    sys.exit(main() or 0)
