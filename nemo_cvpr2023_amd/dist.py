"""Instance-sharded data parallelism for the NeMo fit (one process per GPU, RCCL over xGMI).

The reference is single-process; this is the MI355X-native multi-GPU design of
SURVEY.md 8(e):

* the (instance x frame) batch is sharded BY INSTANCE: rank r owns a contiguous block of videos with
  all their frames, 2-D targets and *private* parameters (camera, instance code, phase network) and
  their Adam state -- no communication for any of those;
* the shared parameters (motion MLP + RBF widths, ~9 MB at h=1000) are replicated; their gradient is
  summed with ONE ``all_reduce`` per Adam step over a single contiguous slice of the flat gradient
  buffer, the loss scalars riding in the last 8 floats of the same slice;
* cross-sample normalisers (number of views present, N) are known on the host because every rank
  draws the same global (view, frame) index stream, so each rank scales its local terms to global
  ones before the reduction (`ShardPlan`); replicas stay bit-identical because they apply the same
  fused Adam to the same reduced buffer.

``backend='nccl'`` is RCCL on ROCm.  xGMI is point-to-point: a 9 MB buffer is latency-, not
bandwidth-bound, so the default is exactly one collective per step and nothing else on the data path.

``shard_mode='buckets'`` (round 3): the shared gradient is reduced in THREE buckets -- [heads + layer 4], [layer 2],
[layer 0 + RBF widths + the 8 loss scalars]: contiguous slices of the flat gradient buffer, laid out in the order the
backward completes them (``engine.ParamLayout.buckets``) -- each issued on a communication stream as soon as its
parameter-gradient GEMMs and bias column sums have run, with the fused Adam of the bucket right behind its collective
on that stream, while the main stream goes on with the rest of the backward.  The step is then three launches (graphs
cut at the bucket boundaries).  What this buys depends on the machine: the MLP backward is the LAST ~15 % of a step, so
at most the first two collectives hide under compute, against two more launch boundaries -- ``bench.py --shard-mode
auto`` times all modes on the box it runs on.

Random minibatches (the published run's mode: batch 512, ``run_scripts_examples/nemomocap-example.sh:17``): a rank's
share of a global draw changes size every step, and a captured HIP graph is per launch size.  Shares are therefore
LAUNCHED at the next multiple of ``pad`` (default 32) samples with masked padding rows (``include/nemo_hip.h``,
nemo_kp_fwd: valid indices, no loss, not counted, exactly-zero gradients) and the per-sample means are taken with
``mr = N_launch / N_global``: two or three graphs serve a whole run at any world size.

Opt-in ``shard_mode='split'`` (``ShardedNemo.set_shard_mode``): a second, 32-byte
all-reduce of the loss scalars is issued on a side stream as soon as they are final (a third of a step before
the gradient is), so the host holds the global losses early and prepares the next launch under the rest of the
backward.  Which of the two is faster depends on the machine's small-message all-reduce latency against the host's
launch latency; ``bench.py --gpus N`` times both on the box it runs on and reports the choice.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .neural_motion_model import NEMO_VERSIONS, ShardInfo, make_init_state


def partition_views(num_views: int, world: int):
    """Contiguous blocks; the first ``num_views % world`` ranks own one extra view."""
    base, extra = divmod(num_views, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((lo, lo + n))
        lo += n
    return out


class ShardPlan:
    """Pure host-side routing of a global batch to one rank (no GPU, unit-tested on CPU)."""

    def __init__(self, num_views, num_frames, rank, world):
        self.V, self.T, self.rank, self.world = num_views, num_frames, rank, world
        self.lo, self.hi = partition_views(num_views, world)[rank]
        self.v_local = self.hi - self.lo

    def full_batch(self):
        """Normalisers of the full-batch step: every view is present on every rank's share."""
        vr = self.v_local / self.V
        return dict(kr=vr, mr=vr, vr=vr, n_global=self.V * self.T)

    def route(self, view_idx, frame_idx):
        """Global (view, frame) draws -> this rank's samples with local view ids + normalisers."""
        view_idx, frame_idx = torch.as_tensor(view_idx).cpu(), torch.as_tensor(frame_idx).cpu()
        mine = (view_idx >= self.lo) & (view_idx < self.hi)
        lv, lf = view_idx[mine] - self.lo, frame_idx[mine]
        n_u_global = int(view_idx.unique().numel())
        n_u_local = int(lv.unique().numel())
        n = int(view_idx.numel())
        return lv, lf, dict(kr=n_u_local / max(n_u_global, 1), mr=int(lv.numel()) / max(n, 1),
                            vr=self.v_local / self.V, n_global=n)


class SequenceSubset:
    """View of a ``multi_view_seqs`` object restricted to a block of videos."""

    def __init__(self, seqs, lo, hi):
        self.num_views, self.num_frames = hi - lo, seqs.num_frames
        self.IMG_D0, self.IMG_D1 = seqs.IMG_D0, seqs.IMG_D1
        self.sequences = seqs.sequences[lo:hi]
        self._parent, self._lo = seqs, lo

    def get_image(self, v, t):
        return self._parent.get_image(v + self._lo, t)


def slice_state(state: dict, lo: int, hi: int) -> dict:
    """Global state_dict -> the slice a rank owning views [lo, hi) holds."""
    out = {}
    for k, v in state.items():
        if k in ('learned_cameras', 'learned_instance_code'):
            out[k] = v[lo:hi]
        elif k.startswith('phase_networks.'):
            i = int(k.split('.')[1])
            if lo <= i < hi:
                out[f'phase_networks.{i - lo}.' + k.split('.', 2)[2]] = v
        else:
            out[k] = v
    return out


class ShardedNemo:
    """``NemoV*`` sharded by instance over the ranks of a process group.  Same call surface for
    the fit (``step / warmup / opt_cam``); indices are GLOBAL view ids."""

    def __init__(self, version, args, multi_view_seqs, device, rank=None, world=None, group=None, seed=0,
                 **assets):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.V, self.T = multi_view_seqs.num_views, multi_view_seqs.num_frames
        if version == 0:
            raise ValueError('instance sharding covers NemoV1..V4 (NemoV0 has no shared motion MLP group)')
        if self.world > self.V:
            raise ValueError(f'{self.world} ranks for {self.V} instances: shard by instance needs world <= V')
        self.plan = ShardPlan(self.V, self.T, self.rank, self.world)
        torch.manual_seed(seed)
        state = make_init_state(args, version, self.V, multi_view_seqs.IMG_D0)    # identical on all ranks
        local = SequenceSubset(multi_view_seqs, self.plan.lo, self.plan.hi)
        self.model = NEMO_VERSIONS[version](args, local, device, **assets)
        # a rank's share of a random minibatch changes size from step to step: launched at multiples of `pad` samples
        self.pad = 32
        self.model.load_state_dict({k: v for k, v in slice_state(state, self.plan.lo, self.plan.hi).items()},
                                   strict=False)
        torch.manual_seed(seed + 1)       # re-synchronise the index stream across ranks
        e = self.model.engine
        a, b = e.layout.span(e.layout.groups['motion'] + e.layout.groups['comm'])
        self._span = (a, b)
        self.args, self.optimizers = args, self.model.optimizers
        # measurement aid (bench.py): False = every collective is skipped, the step is what ONE rank computes
        import atexit, weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())      # (before torch tears the communicator down)
        self.collectives = True
        # RCCL all-reduces can be captured into the step's HIP graph (one launch per sharded step); gloo's cannot
        self.capturable = dist.get_backend(group) == 'nccl'
        self.shard_mode = 'single'
        self.set_shard_mode('single')

    def close(self):
        """Release every captured HIP graph of the model.  Sharded steps capture their RCCL all-reduces into the step's
        graph; those graphs must go BEFORE ``dist.destroy_process_group()`` tears the communicator down."""
        try:
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
            for w in self.model.engine.ws.values():
                w['graphs'].clear()
        except Exception:                  # (interpreter shutdown: best effort)
            pass
        # (deliberately no __del__: a finaliser that synchronises the device can fire from the garbage collector in the
        #  middle of another model's graph capture and invalidate it)

    def set_shard_mode(self, mode):
        """'single': one all-reduce per step (gradient + loss scalars).  'split': + an early 32-byte all-reduce
        of the loss scalars on the side stream.  'buckets': three gradient buckets, each reduced (and Adam-stepped) on a
        communication stream as soon as the backward has completed it.  Must be the same on every rank."""
        if mode not in ('single', 'split', 'buckets'):
            raise ValueError(mode)
        self.shard_mode = mode

    # one collective per Adam step: [d motion MLP | d RBF | loss scalars]
    def _comm(self, engine, update):
        a, b = self._span
        buf = engine.grads[a:b] if update else engine.view('_comm_scalars', engine.grads)
        if self.collectives:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def _comm_small(self, buf):
        if self.collectives:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def _info(self, d, pad=0):
        return ShardInfo(kr=d['kr'], mr=d['mr'], vr=d['vr'], n_global=d['n_global'], comm=self._comm,
                         comm_small=self._comm_small if self.shard_mode == 'split' else None,
                         comm_bucket=self._comm_small if self.shard_mode == 'buckets' else None, pad=pad,
                         capturable=self.capturable, live=self.collectives, comm_log=self._comm_small)

    def _sharder(self):
        """Draw the GLOBAL (view, frame) batch from the CPU RNG (identical on every rank) and keep
        this rank's samples."""
        vi, fi = self.draw_batch()
        lv, lf, d = self.plan.route(vi, fi)
        return lv, lf, self._info(d)

    def step(self, view_idx, frame_idx, update=True, full_batch=False):
        if self.args.batch_size > -1 and not full_batch:
            lv, lf, d = self.plan.route(view_idx, frame_idx)
            return self.model.step(lv, lf, update=update, _shard=self._info(d, pad=self.pad))
        return self.model.step(None, None, update=update, full_batch=True,
                               _shard=self._info(self.plan.full_batch()))

    def warmup(self, steps):
        return self.model.warmup(steps, _sharder=self._sharder)

    def opt_cam(self, steps):
        if self.model.VERSION == 4:
            return self.model.opt_cam(steps, _sharder=self._sharder)
        return self.model.opt_cam(steps, _shard=self._info(self.plan.full_batch()))

    def draw_batch(self):
        B = self.args.batch_size
        return torch.randint(0, self.V, size=(B,)), torch.randint(0, self.T, size=(B,))

    def gather_state_dict(self):
        """Global state_dict assembled on every rank (checkpointing / tests)."""
        local = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        objs = [None] * self.world
        dist.all_gather_object(objs, (self.plan.lo, self.plan.hi, local), group=self.group)
        out = {}
        for lo, hi, sd in sorted(objs, key=lambda t: t[0]):
            for k, v in sd.items():
                if k in ('learned_cameras', 'learned_instance_code'):
                    out.setdefault(k, []).append(v)
                elif k.startswith('phase_networks.'):
                    i = int(k.split('.')[1])
                    out[f'phase_networks.{i + lo}.' + k.split('.', 2)[2]] = v
                else:
                    out[k] = v
        return {k: (torch.cat(v, 0) if isinstance(v, list) else v) for k, v in out.items()}
