"""Drop-in ``NemoV1..V4`` classes on the MI355X fit engine.

Mirrors the public surface of ``nemo/neural_motion_model.py`` that
``scripts/learned_multi_view_recon_nn.py:192-335`` and ``nemo/utils/render_utils.py:90-158`` use:
constructor ``NemoVk(args, multi_view_seqs, device)``, ``step / warmup / opt_cam / get_preds /
get_preds_batch / learned_camera_projection / keypoint_loss / save / load``, attributes
``optimizers, phase_networks, learned_cameras, num_views, num_frames, args, device`` and the
``state_dict`` key names (SURVEY.md section 5 "Checkpoint / resume").

All arithmetic of the fit runs in ``libnemo_hip.so`` through :class:`engine.FitEngine`; the tensors
returned by ``get_preds*`` are plain (non-differentiable) device tensors.

Deliberate deviations from the reference (documented in DESIGN.md):
  * ``loss_dict['kp_loss']`` is the pure 2-D term (CUDA semantics; the CPU backend aliases it);
  * a batch of exactly three samples is NOT special (reference: ``torch.cross`` without ``dim``
    crosses over the batch axis for a (3,3) input, hmr/geometry.py:60);
  * NaN gradients raise ``FloatingPointError`` instead of dropping into ``ipdb`` (:3497-3500);
  * NemoV3/V4 with ``weight_3d_loss == 0`` report ``loss_3d = 0`` (reference: UnboundLocalError).
"""
from __future__ import annotations

import gc
import os
import warnings
from collections import OrderedDict, defaultdict
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, dptr
from .engine import (FOCAL_LENGTH, HEAD_LD, LOSS_TYPES, S_3D, S_GMM, S_KL, S_KP, S_V2V, FitEngine, _stream)

S_INST = 5
S_SMOOTH = 6
S_NAN = 7          # warm-up: number of NaN gradient entries (:3497-3500)


class _Batch:
    """The samples of one step as the kernels see them: indices (host- or device-resident), N = launch size, Nv = real
    samples (N > Nv: masked padding rows), shard normalisers."""
    __slots__ = ('vi', 'fi', 'N', 'Nv', 'padded', 'sh', 'is_full', 'noise', 'host_idx')


class _Plan:
    """How one step is launched (MultiViewModel._plan_step)."""
    __slots__ = ('mode', 'update', 'early', 'graphable', 'segs', 'bgroups', 'in_graph_adam', 'has_inst')


class _LazyInfo(dict):
    """info_dict whose per-step tensors are built on first access."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lazy = {}

    def __missing__(self, key):
        if key in self.lazy:
            self[key] = self.lazy.pop(key)()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.lazy

    def keys(self):
        return list(dict.keys(self)) + list(self.lazy)


class ShardInfo:
    """Normalisers of one rank's share of a step when the (instance x frame) batch is sharded by
    instance (nemo_cvpr2023_amd/dist.py).  kr = n_U_local / n_U_global scales the per-view keypoint
    mean (:3551-3558); mr = N_local / N_global the per-sample means (GMM, KL, v2v, 3-D); vr =
    V_local / V_global the instance-code regulariser.  ``comm(engine)`` all-reduces the shared
    gradients together with the loss scalars.  The single-process default is the identity."""

    def __init__(self, kr=1.0, mr=1.0, vr=1.0, n_global=None, comm=None, comm_small=None, comm_bucket=None, pad=0,
                 capturable=False, live=True, comm_log=None):
        self.kr, self.mr, self.vr, self.n_global, self.comm = kr, mr, vr, n_global, comm
        self.comm_log = comm_log              # all-reduce of a device vector (the loss log of a camera fit, once per phase)
        # False: `comm` / `comm_small` / `comm_bucket` skip their collectives (bench.py's compute-only pass) -- part of the
        # graph key: a step captured with its collectives inside must not be replayed for the other setting
        self.live = bool(live)
        # the collectives may be captured into a HIP graph (RCCL; not gloo, whose device collectives go through the host)
        self.capturable = bool(capturable)
        self.comm_small = comm_small          # all-reduce of a small device tensor (the loss scalars)
        # all-reduce of ONE gradient bucket (a contiguous slice of the flat gradient buffer): when set, update steps
        # reduce the shared gradient in three buckets, each as soon as the backward has completed it (dist.py 'buckets')
        self.comm_bucket = comm_bucket
        # > 0: a minibatch share of n samples is LAUNCHED as ceil(n / pad) * pad samples (masked padding rows, see
        # include/nemo_hip.h nemo_kp_fwd), so that a handful of captured graphs serve every share size
        self.pad = int(pad)


# Seam for harnesses that want to observe or disturb a graph capture (bench.py's multi-GPU fallback tests install their fault
# injection here): called as _capture_hook(sharded, comm_inside) right before a capture opens.  None in the product.
_capture_hook = None


def clip_segments(segs, lo, hi):
    """The parts of Adam launch segments (FusedAdam.segments) that fall inside [lo, hi) of the flat buffer."""
    out = []
    for s_ in segs:
        a, b = max(s_['offset'], lo), min(s_['offset'] + s_['numel'], hi)
        if b > a:
            out.append(dict(s_, offset=a, numel=b - a))
    return out


# ----------------------------------------------------------------------------- parameter holders
class _Linear(nn.Module):
    def __init__(self, weight, bias):
        super().__init__()
        self.weight, self.bias = weight, bias


class _Holder(nn.Module):
    pass


class MonotonicNetwork(nn.Module):
    """monotonic_network.py:7-39.  Parameters are views into the engine's flat buffer; calling the
    module evaluates the warp with the HIP kernel (used by the script to plot, :317-328)."""

    def __init__(self, engine, index, shifts, scales):
        super().__init__()
        self._engine = [engine]
        self.index = index
        self.n_nodes = shifts.numel()
        self.shifts, self.scales = shifts, scales

    def forward(self, x):
        e = self._engine[0]
        x = x.to(e.device, torch.float32).reshape(-1).contiguous()
        n = x.numel()
        out = torch.empty(n + 1, 1, device=e.device)
        ph = torch.empty(n, device=e.device)
        vi = torch.zeros(n, dtype=torch.long, device=e.device)
        check(e.lib.nemo_phase_embed_fwd(n, 1, e.T, e.K, 0, 0, dptr(vi), None, dptr(x),
                                         self.shifts.data_ptr(), self.scales.data_ptr(), 2 * e.K, None, None,
                                         None, 0, dptr(out), 1, dptr(ph), None, None, _stream()), 'nemo_phase_embed_fwd')
        return ph.unsqueeze(1)


class RBF(nn.Module):
    """nemo/rbf.py:11-56 parameter holder (centres buffer + log_sigmas)."""

    def __init__(self, out_features, basis_func, log_sigmas):
        super().__init__()
        self.in_features, self.out_features = 1, out_features
        self.basis_func_name = basis_func
        self.register_buffer('centres', torch.linspace(0, 1, out_features).unsqueeze(1))
        self.log_sigmas = log_sigmas


# ----------------------------------------------------------------------------- optimiser
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam/AdamW-compatible facade over ``nemo_adam_step``.

    ``param_groups[0]['lr']`` is live (ReduceLROnPlateau and the script's LR logging work),
    ``state_dict()/load_state_dict()`` use torch's Adam format.  Step counts are tracked per tensor
    because torch skips tensors whose ``.grad`` is None (e.g. ``linear_out`` during warm-up)."""

    def __init__(self, engine, names, params, lr, weight_decay=0.0, adamw=False, exp_avg=None,
                 exp_avg_sq=None):
        defaults = dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._engine = engine
        self.names = list(names)
        self.adamw = adamw
        self.steps = {n: 0 for n in names}
        self._m = engine.exp_avg if exp_avg is None else exp_avg
        self._v = engine.exp_avg_sq if exp_avg_sq is None else exp_avg_sq
        # (offset, numel) per tensor in MEMORY order -- this runs on the host every step, keep it cheap
        geom = [(n,) + (engine.layout.entries[n][0], int(np.prod(engine.layout.entries[n][1]))) for n in names]
        self._geom = sorted(geom, key=lambda t: t[1])

    def segments(self, active=None):
        """Advance the step counters of the active tensors and return merged launch segments."""
        g = self.param_groups[0]
        lr, wd, steps = float(g['lr']), float(g['weight_decay']), self.steps
        raw = []
        for n, off, numel in self._geom:
            if active is not None and n not in active:
                continue
            steps[n] += 1
            raw.append(dict(offset=off, numel=numel, lr=lr, wd=wd, adamw=self.adamw, step=steps[n]))
        segs = []
        for s in raw:                                          # memory order: adjacent tensors merge
            gap = s['offset'] - (segs[-1]['offset'] + segs[-1]['numel']) if segs else -1
            if 0 <= gap < 4 and segs[-1]['step'] == s['step']:      # (gap: the layout's 16-byte alignment pad)
                segs[-1]['numel'] += gap + s['numel']
            else:
                segs.append(s)
        return segs

    @torch.no_grad()
    def step(self, closure=None, active=None):
        self._engine.adam(self.segments(active), self._m, self._v)

    def zero_grad(self, set_to_none=False):
        a, b = self._engine.layout.span(self.names)
        self._engine.grads[a:b].zero_()

    def _sync_state(self):
        self.state.clear()
        for n, p in zip(self.names, self.param_groups[0]['params']):
            if self.steps[n] > 0:
                self.state[p] = {'step': torch.tensor(float(self.steps[n])),
                                 'exp_avg': self._engine.view(n, self._m),
                                 'exp_avg_sq': self._engine.view(n, self._v)}

    def state_dict(self):
        self._sync_state()
        return super().state_dict()

    def load_state_dict(self, sd):
        sd = deepcopy(sd)
        super().load_state_dict(sd)
        for n, p in zip(self.names, self.param_groups[0]['params']):
            st = self.state.get(p)
            if st:
                self.steps[n] = int(float(st['step']))
                self._engine.view(n, self._m).copy_(st['exp_avg'].to(self._m.device))
                self._engine.view(n, self._v).copy_(st['exp_avg_sq'].to(self._v.device))
            else:
                self.steps[n] = 0
        self._sync_state()


# ----------------------------------------------------------------------------- model
class MultiViewModel(nn.Module):
    """Common part of nemo/neural_motion_model.py:151-280, :2758-3124."""

    VERSION = 1

    def __init__(self, args, multi_view_seqs, device, smpl_assets=None, vposer_state=None, gmm=None):
        super().__init__()
        out_dir = getattr(args, 'out_dir', None)
        args, multi_view_seqs = self._saved_config(args, multi_view_seqs)         # :155-192
        self.args = args
        if not hasattr(self.args, 'include_vs'):
            self.args.include_vs = False
            self.args.include_pare = False
        self.FOCAL_LENGTH = FOCAL_LENGTH
        self.IMG_D0, self.IMG_D1 = multi_view_seqs.IMG_D0, multi_view_seqs.IMG_D1
        self.n_joints = 23
        self.device = torch.device(device)
        self.multi_view_seqs = multi_view_seqs
        self.num_views, self.num_frames = multi_view_seqs.num_views, multi_view_seqs.num_frames
        if out_dir and getattr(args, 'write_config', True):                 # :199-202 (the NEW run's out_dir, :190)
            try:
                import joblib
                os.makedirs(out_dir, exist_ok=True)
                joblib.dump({'args': self.args}, os.path.join(out_dir, 'model_config.p'))
            except Exception:
                pass
        if smpl_assets is None or vposer_state is None or gmm is None:
            from .assets import load_real_assets
            smpl_assets, vposer_state, gmm = load_real_assets(smpl_assets, vposer_state, gmm)
        points, size = collate_gt_2d(multi_view_seqs, args.label_type,
                                     getattr(args, 'label_intersection_threshold', 30.0))
        pose = torch.tensor(np.array([np.array(multi_view_seqs.sequences[v]['pose'])
                                      for v in range(self.num_views)])).float()
        self._engine = [FitEngine(self.VERSION, args, self.num_views, self.num_frames, self.IMG_D0,
                                  self.IMG_D1, smpl_assets, vposer_state, gmm, self.device, points, size,
                                  pose[..., 3:-1].contiguous(), pose[..., -1:].contiguous())]
        e = self.engine
        self.points2d_gt_all, self.gt_bbox_size = e.targets, e.gt_size
        self.hmr_theta, self.hmr_mask = e.hmr_theta, e.hmr_mask
        self.training = False
        # capture each (batch size, mode) variant of the step as a HIP graph after one eager run
        self.use_graphs = os.environ.get('NEMO_GRAPHS', '1') != '0'
        self.GRAPH_AFTER = 1             # eager runs of a (batch size, mode) variant before it is captured
        # sharded steps: collectives inside the graph (NEMO_GRAPH_COMM=0: the graph-then-eager-collectives structure --
        # what bench.py's multi-GPU launcher falls back to when a capture with RCCL inside fails on the machine)
        self.graph_comm = os.environ.get('NEMO_GRAPH_COMM', '1') != '0'
        self.launch_stats = {'replayed': 0, 'other': 0}     # launches that replayed an existing graph / ran eagerly or captured
        self._build_parameters()
        self._init_parameters()
        self._build_optimizers()

    @property
    def engine(self) -> FitEngine:
        return self._engine[0]

    @staticmethod
    def _saved_config(args, multi_view_seqs):
        """:155-192: with ``args.load_ckpt_path = <run>/ckpt/sd_xxxxxx.pt`` the model is built from the ``args`` the
        run saved in ``<run>/model_config.p`` (what ``--test`` / resumed runs of the script rely on, scripts:313-314),
        and the sequences are re-read for THOSE arguments: the 'generic' loader through this package's data layer
        (``include_vs / include_pare`` set as the reference does); 'penn_action' / 'demo' loaders are outside this
        package, so the sequences the caller passed are kept.  No config file: the arguments as given (:186-187)."""
        path = getattr(args, 'load_ckpt_path', '') or ''
        if not path:
            return args, multi_view_seqs
        base = os.path.dirname(os.path.dirname(path))
        cfg_path = os.path.join(base, 'model_config.p')
        if not os.path.exists(cfg_path):
            print('Cannot find saved config .... ')
            return args, multi_view_seqs
        import joblib
        saved = joblib.load(cfg_path)['args']
        kind = getattr(saved, 'data_loader_type', None)
        if kind == 'generic':
            saved.include_vs, saved.include_pare = True, True
            if getattr(saved, 'nemo_cfg', None) is not None:
                from .multi_view_sequence import load_nemo_mocap
                try:
                    multi_view_seqs = load_nemo_mocap(saved.nemo_cfg, saved.start_phase, saved.n_frames)
                except (OSError, KeyError) as ex:        # the data of the saved run is not on this machine
                    warnings.warn(f'saved config: could not re-read its sequences ({ex!r}); using the ones passed in')
        elif kind == 'penn_action':
            saved.include_vs, saved.include_pare = True, False
        elif kind is not None and kind != 'demo':
            raise ValueError('Unsupported `data_loader_type`.')
        return saved, multi_view_seqs

    # ------------------------------------------------------------------ evaluation / rendering surface of the script
    def eval_2d(self, out_dir, num_frames=-1, num_views=-1, view_idxs=[]):
        """:522-710 (called by scripts/learned_multi_view_recon_nn.py:333): writes ``<out_dir>/eval_2d.csv``."""
        from . import evaluation
        return evaluation.eval_2d(self, out_dir, num_frames, num_views, list(view_idxs))

    def eval_3d(self, out_dir, num_frames=-1, num_views=-1, view_idxs=[], dynamic_only=False):
        """:1056-1282 (scripts:334-335): ``eval_3d.csv`` / ``eval_3d_dynamic.csv``."""
        from . import evaluation
        return evaluation.eval_3d(self, out_dir, num_frames, num_views, list(view_idxs), dynamic_only)

    def render_rollout_keypoint_figure(self, fpath=None, *a, **k):
        """:422-520 draws key points over video frames with matplotlib / OpenCV.  Rendering is outside this package
        (out of scope), but the script calls this one unconditionally right after building the model
        (scripts:199-202), so it warns and returns instead of failing the run."""
        warnings.warn('render_rollout_keypoint_figure: rendering is not part of nemo_cvpr2023_amd (nothing written to '
                      f'{fpath!r}); use the reference class on get_preds() outputs for figures')
        return None

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith('render_'):
                def _no_render(*a, **k):
                    raise NotImplementedError(
                        f'{type(self).__name__}.{name}: rendering is out of scope of nemo_cvpr2023_amd -- run the script '
                        'with --render_rollout_figure / --render_video off (scripts/learned_multi_view_recon_nn.py:242, '
                        ':284, :331), or render get_preds() outputs with the reference class')
                return _no_render
            raise

    # ------------------------------------------------------------------ parameters
    def _param(self, name):
        p = nn.Parameter(self.engine.view(name))
        p.grad = self.engine.view(name, self.engine.grads)
        return p

    def _build_parameters(self):
        e = self.engine
        self.learned_cameras = self._param('learned_cameras')
        if e.C > 0:
            self.learned_instance_code = self._param('learned_instance_code')
        lin = lambda name: _Linear(self._param(name + '.weight'), self._param(name + '.bias'))
        if self.VERSION == 0:                                   # RotNet x 2 + FCNN (:58-104, :3148-3162)
            for net in ('learned_poses', 'learned_orient'):
                rn = _Holder()
                rn.net = _Holder()
                rn.net.net = _Holder()
                for k in ('0', '2', '4'):
                    rn.net.net.add_module(k, lin(f'{net}.net.net.{k}'))
                rn.linear = lin(f'{net}.linear')
                setattr(self, net, rn)
            tn = _Holder()
            tn.net = _Holder()
            for k in ('0', '2', '4'):
                tn.net.add_module(k, lin(f'learned_trans.net.{k}'))
            self.learned_trans = tn
        else:
            lm = _Holder()
            lm.net = _Holder()
            lm.net.net = _Holder()
            for k in ('0', '2', '4'):
                lm.net.net.add_module(k, lin(f'learned_motion.net.net.{k}'))
            lm.rot_out = lin('learned_motion.rot_out')
            lm.linear_out = lin('learned_motion.linear_out')
            self.learned_motion = lm
        self.learned_betas = nn.Parameter(e.betas)
        self.phase_networks = nn.ModuleList([
            MonotonicNetwork(e, i, self._param(f'phase_networks.{i}.shifts'),
                             self._param(f'phase_networks.{i}.scales')) for i in range(e.V)])
        if e.D > 0:
            self.phase_rbf = RBF(e.D, self.args.rbf_kernel, self._param('phase_rbf.log_sigmas'))

    @torch.no_grad()
    def _init_parameters(self):
        e = self.engine
        st = make_init_state(self.args, self.VERSION, e.V, self.IMG_D0)
        for k, v in st.items():
            e.view(k).copy_(v)

    def _build_optimizers(self):
        e, a, G = self.engine, self.args, self.engine.layout.groups
        named = dict(self.named_parameters())
        mk = lambda grp, lr, wd=0.0, adamw=False: FusedAdam(e, G[grp], [named[n] for n in G[grp]], lr, wd,
                                                            adamw)
        self.opt_cameras = mk('cameras', a.lr_camera)
        self.opt_phase = mk('phase', a.lr_phase)
        if self.VERSION == 0:                                   # :3171-3198
            aw = a.opt_human == 'adamw'
            self.opt_poses = mk('poses', a.lr_pose, a.wd_human, aw)
            self.opt_orient = mk('orient', a.lr_orient, a.wd_human, aw)
            self.opt_trans = mk('trans', a.lr_trans)
            self.optimizers = [self.opt_cameras, self.opt_poses, self.opt_orient, self.opt_trans, self.opt_phase]
        else:
            self.opt_motion = mk('motion', a.lr_human, a.wd_human, a.opt_human == 'adamw')
            self.optimizers = [self.opt_cameras, self.opt_motion, self.opt_phase]
        if e.C > 0:
            self.opt_instance = mk('instance', a.lr_instance)
            self.optimizers.append(self.opt_instance)
        self.schedulers = []
        if a.lr_factor < 1:
            self.schedulers = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, factor=a.lr_factor, min_lr=1e-6)
                               for o in self.optimizers]

    def _adam_all(self, optimizers, active=None):
        segs = []
        for o in optimizers:
            segs += o.segments(active)
        self.engine.adam(segs)

    # ------------------------------------------------------------------ checkpointing (:257-280)
    def save(self, path):
        torch.save({'model_sd': self.state_dict(), 'opt_sd': [o.state_dict() for o in self.optimizers]}, path)

    def load(self, path):
        saved = torch.load(path, map_location=self.device, weights_only=False)
        sd = {k: v for k, v in saved['model_sd'].items()
              if not k.startswith(('vp', 'pose_prior', 'renderer', 'smpl'))}
        self.load_state_dict(sd, strict=False)
        for i, opt in enumerate(self.optimizers):
            opt.load_state_dict(saved['opt_sd'][i])

    # ------------------------------------------------------------------ helpers
    def _idx(self, t):
        return torch.as_tensor(t).to(self.device, torch.long).contiguous()

    def full_indices(self):
        if getattr(self, '_full_idx', None) is None:        # constant: built once, not per step
            v = torch.arange(self.num_views, device=self.device).repeat_interleave(self.num_frames)
            f = torch.arange(self.num_frames, device=self.device).repeat(self.num_views)
            self._full_idx = (v, f)
        return self._full_idx

    def frame_idx_to_raw_phase(self, frame_idx):
        return torch.linspace(0, 1, self.num_frames).to(frame_idx.device)[frame_idx]      # :2978-2984

    def keypoint_loss(self, pred, gt, weight, gt_size=None, loss_type=None):
        """:2806-2843, elementwise API-parity helper (the fit uses the fused kernel)."""
        loss_type = loss_type or self.args.loss
        m = (weight > 0.5).float()
        rho2 = 100.0 ** 2
        gm = lambda r2: rho2 * (r2 / (r2 + rho2))
        if loss_type in ('rmse_resized', 'mse_robust_resized'):
            s = gt_size.unsqueeze(-1).unsqueeze(-1)
            k = 1000.0 if loss_type == 'mse_robust_resized' else 1.0
            pred, gt = pred / s * k, gt / s * k
        if loss_type in ('rmse', 'rmse_resized'):
            return m * torch.sqrt(1e-6 + ((pred - gt) ** 2).sum(-1, keepdim=True))
        if loss_type == 'mse':
            return m * (pred - gt) ** 2
        if loss_type == 'rmse_robust':
            return m * gm(torch.sqrt(((pred - gt) ** 2).sum(-1)).unsqueeze(-1))
        if loss_type in ('mse_robust', 'mse_robust_resized'):
            return m * gm((pred - gt) ** 2)
        raise ValueError(loss_type)

    def learned_camera_projection(self, input_points3d, view_idx):
        """:3073-3124 for arbitrary (N, J, 3) points."""
        e = self.engine
        pts = input_points3d.detach().to(self.device, torch.float32).contiguous()
        vi = self._idx(view_idx)
        out = torch.empty(pts.shape[0], pts.shape[1], 2, device=self.device)
        check(e.lib.nemo_project(pts.shape[0], pts.shape[1], e.V, dptr(pts), dptr(vi), e.p('learned_cameras'),
                                 FOCAL_LENGTH, e.cx, e.cy, dptr(out), _stream()), 'nemo_project')
        return out

    # ------------------------------------------------------------------ predictions
    def _noise(self, N):
        a, e = self.args, self.engine
        if self.VERSION >= 3 and self.training and getattr(a, 'code_noise', 0) > 0 and e.C > 0:
            return a.code_noise * torch.randn(N, e.C, device=self.device)
        return None

    @torch.no_grad()
    def get_preds_batch(self, view_idx, frame_idx, add_trans=True, phases=None, detach_pose=False,
                        with_vertices=True):
        """:3637-3672 / :3968-4008.  Returns fresh tensors: v (N,NV,3), j (N,25,3), poses (N,69),
        orient (N,6), orient_aa (N,3), trans (N,3)."""
        e = self.engine
        vi = self._idx(view_idx)
        fi = self._idx(frame_idx) if frame_idx is not None else None
        N = vi.numel()
        w = e._ws(N)
        raw = None if phases is None else phases.to(self.device, torch.float32).reshape(-1).contiguous()
        e.forward_pose(w, N, vi, fi, raw_phase=raw, code_noise=self._noise(N), train=False)
        j3d = torch.empty(N, e.ctx.n_out, 3, device=self.device)
        e.forward_joints(w, N, vi, fi, with_loss=False, add_trans=add_trans, j3d=j3d)
        trans = w['TR'][:N] - w['TR'][N:N + 1] if not e.start_global_traj_anywhere else w['TR'][:N].clone()
        out = {'view_idx': view_idx, 'frame_idx': frame_idx, 'j': j3d, 'poses': w['AA'][:N, 3:].clone(),
               'orient': w['ROT'][:N, :6].clone(), 'orient_aa': w['AA'][:N, :3].clone(), 'trans': trans}
        if with_vertices:
            out['v'] = self._vertices(w, N, trans if add_trans else None)
        return out

    def _vertices(self, w, N, trans):
        e = self.engine
        NV3, ldP = 3 * e.NV, e.ctx.ldP
        verts = torch.empty(N, e.NV, 3, device=self.device)
        chunk = 4096
        VP = torch.empty(min(N, chunk), ldP, device=self.device)
        for c0 in range(0, N, chunk):
            n = min(chunk, N - c0)
            e.gemm(0, 0, n, NV3, 207, w['PF'].data_ptr() + 4 * c0 * 208, 208, e.ctx.posedirs, ldP, dptr(VP),
                   ldP, bias=e.ctx.v_shaped)
            t = None if trans is None else trans[c0:c0 + n].contiguous()
            check(e.lib.nemo_skin_vertices(e.ctx.handle, n, dptr(VP), ldP, w['A'].data_ptr() + 4 * c0 * 288,
                                           dptr(t), 3, verts.data_ptr() + 4 * c0 * NV3, _stream()),
                  'nemo_skin_vertices')
        return verts

    def get_preds(self, add_trans=True):
        """:2986-3003: all (view, frame) pairs, tensors shaped (V, T, ...)."""
        vi, fi = self.full_indices()
        p = self.get_preds_batch(vi, fi, add_trans=add_trans)
        V, T = self.num_views, self.num_frames
        return {k: v.reshape(V, T, *v.shape[1:]) for k, v in p.items()}

    # ------------------------------------------------------------------ the hot path
    def _forward_backward(self, w, N, vi, fi, update, use_vposer=True, detach_pose=False, sh=None,
                          smooth_ok=False, extra_losses=None, publish=False, part='all', adam_segs=0, use_gmm=None):
        """Forward of :3511-3584 (+V3 extras) and, when ``update``, the whole backward down to the
        parameter gradients.  After the pose MLP the step forks into three branches that run CONCURRENTLY on three HIP
        streams (most of their kernels are too small to fill 256 CUs):
          main : FK -> pre-contracted joints -> projection -> 2-D loss -> its backward (dR, dTR, dcams)
          side : VPoser encode -> decode -> rotations + FK of both mesh bodies (the mesh term waits for it)
          side2: everything that accumulates into dAA -- GMM prior, 3-D pose term, then, once the encoder output exists,
                 KL and its backward through the frozen encoder (one stream for all of them: their `+=` into dAA are plain
                 read-modify-writes) -- and the key-point loss scalar
        joins, runs the full-mesh v2v term alone, then the rot6d / MLP backward.  (Schedules measured and dropped --
        priors on the VPoser stream, late joins, the loss hand-over on the main chain: profiles/r0{2,3}_experiments.md.)"""
        e, a = self.engine, self.args
        sh = sh or ShardInfo()
        use_gmm = use_vposer if use_gmm is None else use_gmm
        e.scal = w['scal']               # loss-scalar slots of this workspace (part of the arena)
        # part: 'all', or the two halves a sharded step launches separately -- 'head' = everything up to the
        # point where the loss scalars are final, 'tail' = the rest of the backward (see step())
        if part == 'tail':
            return self._backward_tail(w, N, vi, fi, update, use_vposer, sh)
        if part in ('k1', 'k2'):             # bucketed sharded step: the later stages of the MLP backward (see step())
            return e.backward_mlp(w, N, vi, fi, None, stages=(int(part[1]),), bucketed=True)
        # first launch: phase / RBF / code forward; its further blocks zero the loss scalars, view accumulators, dAA, dJp,
        # dA2, dPF2 and the gradient buffer and advance the device Adam table (nemo_phase_embed_fwd_begin)
        e.forward_pose(w, N, vi, fi, code_noise=self._noise(N), train=bool(update),
                       begin=(w['zero_arena'], bool(update), adam_segs))
        main = torch.cuda.current_stream()
        side, side2 = e.side_stream, e.side_stream2
        pose_done = main.record_event()
        aa69, daa69 = w['AA'].data_ptr() + 12, w['dAA'].data_ptr() + 12
        # The main-stream branch (the longest of the three) is ENQUEUED FIRST: a replayed HIP graph keeps a node's first
        # successor on the hardware queue of the node and starts the others on further queues behind a cross-queue
        # barrier that costs 15 - 30 us (profiles/r02_kernel_trace_v1.md) -- that delay has to land on the short branches.
        w_s = float(getattr(a, 'weight_smooth', 0) or 0)
        smooth = bool(w_s and N == e.V * e.T and smooth_ok)
        if update and e.view_cnt is not None and not smooth:
            # projection, 2-D loss and their backward in ONE launch: the per-view normaliser only needs the per-view sample
            # counts, which the host knows from the indices it staged (nemo_kp_fwd_bwd)
            Mq = e.joint_functionals(w, N)
            e.backward_kp(w, N, vi, fi, Mq, mean_mode=0, upstream=float(sh.kr), detach_pose=detach_pose,
                          fused_counts=e.view_cnt)
            # the loss scalar's kernel (side2, below) only needs the view accumulators -- complete behind the fused launch -- not the
            # joint-functional adjoint and the FK backward behind it: waiting for those put a 3 us kernel and two cross-queue hops
            # between the end of the longest branch and the mesh kernel
            kp_done = e.kp_acc_done if os.environ.get('NEMO_KP_FIN_LATE') != '1' else main.record_event()
        else:
            Mq = e.forward_joints(w, N, vi, fi, with_loss=True, mean_mode=0, finalize=False)
            kp_done = main.record_event()        # (view accumulators complete: the loss scalar is finalised on side2, below)
            # optional temporal smoothness of the output joints (not part of the published step; only defined on
            # complete (view, frame) sequences, i.e. full-batch steps -- see DESIGN.md section 8, row f-4)
            dj = None
            if smooth:
                check(e.lib.nemo_smooth_fwd_bwd(e.V, e.T, e.ctx.n_out, dptr(w['j3d']), w_s,
                                                e.scal.data_ptr() + 4 * S_SMOOTH, dptr(w['dj3d']) if update else None,
                                                _stream()), 'nemo_smooth_fwd_bwd')
                dj = w['dj3d'] if update else None
            if update:
                e.backward_kp(w, N, vi, fi, Mq, mean_mode=0, upstream=float(sh.kr), detach_pose=detach_pose,
                              dj3d_extra=dj, norm_from_acc=True)
        side.wait_event(pose_done)
        side2.wait_event(pose_done)
        enc_done = None
        with torch.cuda.stream(side):
            if use_vposer:
                # the decoder starts from the encoder's hidden activation (composed first layer); the (mu | logvar) product
                # runs on side2 in front of the KL term; 6-D -> axis-angle of the decoder output inside v2v_prep
                enc_done = e.forward_vposer(w, N)          # always evaluated, :3569
                e.forward_v2v_pre(w, N)   # rotations + FK of both mesh bodies: only the poses are needed
        with torch.cuda.stream(side2):
            st = _stream()
            if use_gmm:
                g = e.gmm
                check(e.lib.nemo_gmm_fwd_bwd(N, g['M'], 69, aa69, 72, dptr(g['means']), dptr(g['prec']),
                                             dptr(g['log_nllw']), dptr(w['gmm_ws']),
                                             e.scal.data_ptr() + 4 * S_GMM, None,
                                             float(a.weight_gmm_loss) * sh.mr,
                                             daa69 if (update and a.weight_gmm_loss) else None, 72, e.nvalid, st),
                      'nemo_gmm_fwd_bwd')
            if self.VERSION >= 3 and getattr(a, 'weight_3d_loss', 0):
                check(e.lib.nemo_pose3d_fwd_bwd(N, 69, aa69, 72, dptr(e.hmr_theta), dptr(e.hmr_mask),
                                                dptr(vi), dptr(fi), e.T, e.scal.data_ptr() + 4 * S_3D,
                                                float(a.weight_3d_loss) * sh.mr, daa69 if update else None,
                                                72, e.nvalid, st), 'nemo_pose3d_fwd_bwd')
            if use_vposer:
                side2.wait_event(enc_done)
                e.vposer_mulv(w, N)
                e.vposer_kl(w, N)
                if update and a.weight_vp_z_loss:
                    e.backward_vposer_kl(w, N, float(a.weight_vp_z_loss) * sh.mr)
            side2.wait_event(kp_done)
            e.finalize_kp(w, mean_mode=0)
        main.wait_stream(side)
        main.wait_stream(side2)
        # the fused mesh kernel is sized to fill the machine in exactly one resident wave of blocks:
        # it runs alone (anything co-scheduled pushes part of its grid into a second wave, +60 %)
        if extra_losses is not None:
            extra_losses()
        # Every loss scalar is final once the mesh kernel has run: hand them to the host from there (nemo_publish_scalars)
        # while the backward goes on.  When the whole backward follows in this launch, the hand-over leaves the main
        # chain: a branch on side2 that forks right behind the mesh kernel -- before the blend-shape adjoint -- and is
        # enqueued LAST (a replayed graph keeps the first-enqueued successor on the queue, see above), joined before Adam.
        # Measured (same box, three runs each): 0.494 ms at one instance / 1.557 ms at C2 against 0.512 / 1.576 with the
        # hand-over on the main stream, before or after the adjoint -- its system-scope release stalls the queue it is on.
        loss_final = []
        pub_side = False
        pub_aside = bool(publish and update and part == 'all' and use_vposer)
        if use_vposer:
            e.forward_v2v(w, N, need_grad=bool(update and a.weight_vp_loss), pre_done=True,
                          after_loss=(lambda: loss_final.append(main.record_event())) if pub_aside else None)
        if publish and not pub_aside:
            e.publish_scalars()
        if pub_aside and N + 1 <= e.SMALL_BATCH_ROWS:
            # Small batches (minibatch steps, a rank's share): the hand-over goes onto the VPoser stream, which is idle by
            # now, enqueued right here.  Enqueued last on side2 (below) it shares a hardware queue with the grouped
            # parameter-gradient launch of the small-batch backward and reaches the host only at the END of the step
            # (profiles/r04_kernel_trace_phases.md) -- the host then prepares the next minibatch on an idle GPU: 0.625 ->
            # 0.596 ms per minibatch-512 step, 0.633 -> 0.620 at two instances; no difference at 8 x 300.
            side.wait_event(loss_final[0])
            with torch.cuda.stream(side):
                e.publish_scalars()
            pub_aside, pub_side = False, True
        if not update or part == 'head':
            return
        self._backward_tail(w, N, vi, fi, update, use_vposer, sh, stages=(0,) if part == 'k0' else (0, 1, 2),
                            bucketed=part == 'k0')
        if pub_side:
            main.wait_stream(side)            # (every forked stream rejoins the launch's stream)
        if pub_aside:
            side2.wait_event(loss_final[0])
            with torch.cuda.stream(side2):
                e.publish_scalars()
            main.wait_stream(side2)

    def _backward_tail(self, w, N, vi, fi, update, use_vposer, sh, stages=(0, 1, 2), bucketed=False):
        e, a = self.engine, self.args
        st = _stream()
        # v2v_prep_bwd + rot6d_bwd + the trans_0 row sum in one launch
        v2v = bool(use_vposer and a.weight_vp_loss)
        anchored = not e.start_global_traj_anywhere
        check(e.lib.nemo_pose_bwd_fused(
            N, dptr(w['ROT']), HEAD_LD, 1, dptr(w['dR']), dptr(w['dAA']), dptr(w['dROT']), HEAD_LD,
            dptr(w['AA']) if v2v else None, dptr(w['dR2']) if v2v else None,
            float(a.weight_vp_loss) * sh.mr / float(N * e.NV * 3) if v2v else 0.0,
            dptr(w['dTR']) if anchored else None, HEAD_LD, 1, e.head_meta(w, anchored), st), 'nemo_pose_bwd_fused')
        if not anchored:
            e.finish_trans_grad(w, N)
        e.backward_mlp(w, N, vi, fi, None, stages=stages, bucketed=bucketed)

    # ---- launch machinery shared by step / warmup / opt_cam ----------------------------------------------------------
    def _captured(self, w, key, fn, sharded=False, comm_inside=False):
        """Run ``fn()`` -- a closure that only ENQUEUES device work on the current stream(s), no host read-back -- as a
        replayed HIP graph: the first sight of ``key`` in workspace ``w`` runs eagerly (sets kernel attributes, sizes
        pools; a shape that shows up once -- a rank's share of a random minibatch -- is not worth a capture), the second
        is captured, every later one replayed.  ``comm_inside``: fn contains RCCL collectives; if their capture fails this
        model goes back to the graph-then-eager-collectives structure from the next step on (``self.graph_comm``)."""
        entry = w['graphs'].get(key)
        self.launch_stats['replayed' if isinstance(entry, torch.cuda.CUDAGraph) else 'other'] += 1
        e = self.engine
        if isinstance(entry, torch.cuda.CUDAGraph):
            # weight records of the split-precision chain (engine.wrec_ok): a graph captured while they were fresh has no pass over
            # the weights in front of its forward -- if they are stale now (another launch updated the weights without refreshing them,
            # a checkpoint was loaded) that pass runs here, eagerly, in front of the replay; afterwards the flag is what the captured
            # body left it as
            pre, post = w['graph_wrec'][key]
            if pre and not e.wrec_ok():
                e.refresh_weight_records(w)
            entry.replay()
            e._wrec_ok, e._wrec_version = post, e.params._version
            return
        if entry == 'eager':
            return fn()
        if not isinstance(entry, torch.cuda.CUDAGraph) and (entry or 0) < self.GRAPH_AFTER:
            w['graphs'][key] = (entry or 0) + 1
            return fn()
        wrec_pre = e.wrec_ok()
        if not isinstance(entry, torch.cuda.CUDAGraph):           # capture the launches of `fn` as one HIP graph
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # The cyclic garbage collector stays off while the capture is open (torch.cuda.graph collects once
            # BEFORE it begins): a finaliser that touches the device -- a stale graph, a communicator kept alive by
            # one -- firing between two captured launches aborts the process.
            gc_on = gc.isenabled()
            gc.disable()
            try:
                if _capture_hook is not None:
                    _capture_hook(sharded, comm_inside)
                # (a process group's watchdog thread polls events of earlier collectives -- legal next to a capture in
                #  'thread_local' mode, an invalidated capture in the default 'global' mode.  Any capture of a process
                #  that HAS a device process group takes that mode, sharded launch or not: warm-up and camera-fit
                #  iterations of a multi-GPU run are captured too)
                pg = torch.distributed.is_available() and torch.distributed.is_initialized()
                with torch.cuda.graph(g, capture_error_mode='thread_local' if (sharded or pg) else 'global'):
                    fn()
            except RuntimeError as ex:
                # e.g. a capture invalidated by another thread of the process (a collective's
                # watchdog): nothing of fn has run; keep launching this variant kernel by
                # kernel -- same HIP kernels, only the launch overhead comes back
                warnings.warn(f'HIP graph capture failed ({ex}); this step variant runs un-captured')
                torch.cuda.synchronize()
                w['graphs'][key] = 'eager'
                if comm_inside:
                    # a collective that would not capture: from the next step on this model goes back to a graph of
                    # the step followed by eager collectives / hand-over / Adam (the same collectives in the same
                    # order, so ranks that did capture stay in step with this one)
                    self.graph_comm = False
                return fn()
            else:
                w['graphs'][key] = entry = g
                w.setdefault('graph_wrec', {})[key] = (wrec_pre, e._wrec_ok)
            finally:
                if gc_on:
                    gc.enable()
        entry.replay()

    def _stage_indices(self, w, vi, fi, N, n_valid=None, full=False):
        """Step inputs of a captured launch live in the workspace's device-resident index buffers.  Host-resident indices
        go through pinned staging: a pageable H2D copy would block the host until the previous step -- still running,
        the losses are handed over mid-step -- has drained.  The staging buffer is free: the previous copy out of it was
        enqueued before the launch whose losses we already hold.  ``n_valid``: the number of real samples of a padded
        launch, stored behind the view indices; behind that the V per-view sample counts of the batch (what
        nemo_kp_fwd_bwd normalises with) whenever the host can know them -- host-resident indices, or a full batch
        (``full``: T samples per view); all of it travels in ONE copy.  Sets ``engine.view_cnt`` accordingly."""
        e = self.engine
        cap, V = w['cap'], e.V
        src = (vi.data_ptr(), fi.data_ptr(), vi._version, fi._version, N, n_valid)
        if w.get('_static_src') != src or vi is not w.get('_static_vi'):   # (full batch: cached, unchanged)
            if vi.device.type == 'cpu' and fi.device.type == 'cpu':
                if '_idx_pin' not in w:
                    w['_idx_pin'] = torch.zeros(2, cap + 1 + V, dtype=w['vi_static'].dtype).pin_memory()
                pin = w['_idx_pin']
                pin[0, :N].copy_(vi)
                pin[1, :N].copy_(fi)
                pin[0, cap] = N if n_valid is None else n_valid
                pin[0, cap + 1:] = torch.bincount(vi[:N if n_valid is None else n_valid], minlength=V)
                w['vi_static'].copy_(pin[0], non_blocking=True)
                w['fi_static'][:N].copy_(pin[1, :N], non_blocking=True)
                w['_cnt_ok'] = True
            else:
                w['vi_static'][:N].copy_(vi)
                w['fi_static'][:N].copy_(fi)
                if n_valid is not None:
                    w['vi_static'][cap:cap + 1].fill_(n_valid)
                w['_cnt_ok'] = bool(full)
                if full:
                    w['vi_static'][cap + 1:].fill_(e.T)
            w['_static_src'], w['_static_vi'] = src, vi
        e.view_cnt = w['vi_static'][cap + 1:].data_ptr() if w.get('_cnt_ok') else None
        return w['vi_static'][:N], w['fi_static'][:N]      # (the workspace may be larger than this batch)

    def _prepare_batch(self, view_idx, frame_idx, full_batch, sh):
        """-> _Batch: the samples of one step as the kernels will see them (indices, launch size, shard normalisers)."""
        e, a = self.engine, self.args
        b = _Batch()
        b.is_full = not (a.batch_size > -1 and not full_batch)
        b.noise = bool(self.VERSION >= 3 and self.training and getattr(a, 'code_noise', 0) > 0)
        b.host_idx = False
        if not b.is_full:
            vi, fi = torch.as_tensor(view_idx), torch.as_tensor(frame_idx)
            # Host-resident indices (what fit.run_fit / draw_batch produce) stay on the host when the step replays a graph
            # (_stage_indices); device-resident ones (the reference script moves them first, scripts:291-296) are used as
            # they are
            b.host_idx = (vi.numel() > 0 and vi.device.type == 'cpu' and fi.device.type == 'cpu'
                          and vi.dtype == torch.long and fi.dtype == torch.long)
            if not (b.host_idx and self.use_graphs and e.timers is None and not b.noise):
                vi, fi = self._idx(vi), self._idx(fi)
        else:
            vi, fi = self.full_indices()
        b.Nv = b.N = vi.numel()              # Nv: real samples; N: samples the kernels are launched with
        # Padded launch (a rank's share of a sharded random minibatch, scripts:291-296 / nemomocap-example.sh:17): the
        # share size changes every step, a captured graph is per launch size -- so the share is rounded up to a multiple
        # of sh.pad with masked rows (valid indices, no loss, no count, zero gradient: include/nemo_hip.h) and the
        # per-sample means are taken over the launch size with mr = N_launch / N_global.
        b.padded = bool(sh.pad and not b.is_full and b.Nv > 0 and sh.n_global)
        if b.padded:
            b.N = -(-b.Nv // sh.pad) * sh.pad
            assert b.N <= 8192, 'padded launches are single-chunk (minibatch shares)'
            vi = torch.nn.functional.pad(vi, (0, b.N - b.Nv))         # (view 0, frame 0): valid memory, masked out
            fi = torch.nn.functional.pad(fi, (0, b.N - b.Nv))
            sh = ShardInfo(kr=sh.kr, mr=b.N / float(sh.n_global), vr=sh.vr, n_global=sh.n_global, comm=sh.comm,
                           comm_small=sh.comm_small, comm_bucket=sh.comm_bucket, pad=sh.pad, capturable=sh.capturable,
                           live=sh.live)
        b.vi, b.fi, b.sh = vi, fi, sh
        return b

    def _plan_step(self, b, update):
        """-> _Plan: HOW this step is launched.  One place decides the launch structure; `_graph_key` is the one place
        that says what a captured launch bakes in.

        mode 'plain'   : one launch (graph) of the whole step [+ Adam]; single GPU, or a sharded step whose collectives
                         cannot be captured (gloo): then graph -> eager all-reduce -> hand-over -> Adam;
             'split'   : two launches ('head' / 'tail'), the 32-byte loss all-reduce + hand-over on the side stream between;
             'buckets' : three launches (k0 / k1 / k2), a gradient bucket reduced + Adam-stepped behind each;
             'allc' / 'splitc' / 'bucketc': the same three structures as ONE captured launch with the RCCL collectives,
                         the hand-over and Adam inside the graph (sharded, nccl backend, self.graph_comm)."""
        e, a, sh = self.engine, self.args, b.sh
        pl = _Plan()
        pl.update = bool(update)
        pl.has_inst = bool(self.VERSION >= 3 and getattr(a, 'weight_instance_loss', 0) and e.C > 0)
        # single-GPU steps hand the losses to the host as soon as they are final (engine.publish_scalars)
        pl.early = bool(e.early_readback and sh.comm is None and b.N > 0)
        bucketed = bool(update and sh.comm is not None and sh.comm_bucket is not None and self.VERSION >= 1)
        split = bool(update and sh.comm is not None and sh.comm_small is not None and e.early_readback and not bucketed)
        pl.graphable = bool(self.use_graphs and b.N > 0 and e.timers is None and not b.noise)
        pl.segs = None
        if update:
            pl.segs = []
            for o in self.optimizers:
                pl.segs += o.segments(None)
        cap_ok = bool(self.graph_comm and sh.capturable and pl.graphable and update and sh.comm is not None
                      and e.early_readback and self.VERSION >= 1)
        pl.bgroups = None
        pl.mode = 'plain'
        if bucketed:
            pl.mode = 'buckets'
            if cap_ok:
                # the device Adam table grouped by bucket: [bucket 0 | bucket 1 | bucket 2 | private], (start, count, max numel)
                bk = e.layout.buckets()
                lo_, hi_ = bk[2][0], bk[0][1]
                groups = [clip_segments(pl.segs, *bk[i]) for i in range(3)]
                groups.append(clip_segments(pl.segs, 0, lo_) + clip_segments(pl.segs, hi_, e.layout.total))
                if sum(len(g_) for g_ in groups) <= _lib.ADAM_MAX_SEG:
                    pl.bgroups, at = [], 0
                    for g_ in groups:
                        pl.bgroups.append((at, len(g_), max([s_['numel'] for s_ in g_], default=0)))
                        at += len(g_)
                    pl.segs = [s_ for g_ in groups for s_ in g_]
                    pl.mode = 'bucketc'
        elif cap_ok:
            pl.mode = 'splitc' if split else 'allc'
        elif split:
            pl.mode = 'split'
        # the fused Adam runs inside the captured graph unless a gradient all-reduce must come first (and stays outside)
        pl.in_graph_adam = bool(update and (sh.comm is None or pl.mode in ('allc', 'splitc', 'bucketc')))
        return pl

    def _graph_key(self, b, pl, part):
        """Everything a captured launch bakes in besides device-resident inputs: the launch size and mode, the shard
        normalisers, the engine switches of the public NemoV2 setters, the loss weights and loss type, and the geometry
        of the in-graph Adam."""
        e, sh = self.engine, b.sh
        return ('step', b.N, pl.update, b.is_full, sh.kr, sh.mr, sh.vr, sh.comm is not None, sh.live, part, pl.early,
                b.padded, e.view_cnt is not None, e.detach_articulation, e.start_global_traj_anywhere, pl.has_inst, self._weights_key(),
                e.ctx.skin_sparse_flag,
                tuple((s_['offset'], s_['numel']) for s_ in pl.segs) if pl.in_graph_adam else None)

    def _inst_term(self, sh, update):                                                 # :3864-3867
        e, a = self.engine, self.args
        n_code = e.V * e.C
        check(e.lib.nemo_sqmean_fwd_bwd(n_code, e.p('learned_instance_code'), e.scal.data_ptr() + 4 * S_INST,
                                        e.g('learned_instance_code') if update else None,
                                        2.0 * float(a.weight_instance_loss) * sh.vr / n_code, _stream()),
              'nemo_sqmean_fwd_bwd')

    def _body(self, w, b, pl, vi_, fi_, adam_table, part='all', run_adam=True):
        """Everything of the step (or of one part of it) that runs on the device without host interaction."""
        e = self.engine
        # full batches are view-major by construction: the phase backward may find a view's samples by a search (engine.batch_sorted).
        # The flag is valid for THIS body only -- it is taken back on every way out, so that no later caller of engine.phase_bwd /
        # backward_mlp inherits it (a stale True gives wrong d_shifts / d_scales for an unsorted batch, silently)
        e.batch_sorted = bool(b.is_full) and not b.padded
        try:
            return self._body_impl(w, b, pl, vi_, fi_, adam_table, part, run_adam)
        finally:
            e.batch_sorted = False

    def _body_impl(self, w, b, pl, vi_, fi_, adam_table, part, run_adam):
        e, sh, update = self.engine, b.sh, pl.update
        if part == 'bucketc':
            # 'buckets' mode as ONE launch: the three gradient buckets are all-reduced on the communication stream as
            # soon as the backward has completed them, each with its share of the fused Adam (device table, segments
            # grouped by bucket) right behind its collective; the main stream goes on with the backward meanwhile
            main, cs = torch.cuda.current_stream(), e.comm_stream
            bk = e.layout.buckets()
            slot = e.view('_comm_scalars', e.grads)
            for i, kpart in enumerate(('k0', 'k1', 'k2')):
                self._body(w, b, pl, vi_, fi_, adam_table if i == 0 else None, kpart, run_adam=False)
                cs.wait_event(main.record_event())
                with torch.cuda.stream(cs):
                    if i == 2:
                        torch.mul(e.scal, self._shard_weights(sh), out=slot)
                    sh.comm_bucket(e.grads[bk[i][0]:bk[i][1]])
                    if i == 2:
                        e.publish_scalars(slot)
                    st_, n_, mx_ = pl.bgroups[i]
                    if n_:
                        e.adam_from_table(n_, mx_, st_)
            st_, n_, mx_ = pl.bgroups[3]                      # this rank's private parameters, on the main stream
            if n_:
                e.adam_from_table(n_, mx_, st_)
            main.wait_stream(cs)
            return
        if part in ('allc', 'splitc'):
            # Sharded update step with its collectives INSIDE the launch (one captured graph per step: RCCL all-reduces
            # are capturable, tools/debug/graph_capture_rccl.py): body -> weighted loss scalars -> all-reduce(s) ->
            # hand-over to the host -> Adam.  'splitc': the 32-byte all-reduce of the loss scalars and their hand-over
            # fork to the side stream after the first half, as in ShardedNemo's 'split' mode.
            main = torch.cuda.current_stream()
            if part == 'splitc':
                self._body(w, b, pl, vi_, fi_, adam_table, 'head', run_adam=False)
                self._reduce_scalars_on_side_stream(sh)
                if b.N > 0:
                    self._body(w, b, pl, vi_, fi_, None, 'tail', run_adam=False)
                sh.comm(e, update)
            else:
                self._body(w, b, pl, vi_, fi_, adam_table, 'all', run_adam=False)
                slot = e.view('_comm_scalars', e.grads)
                torch.mul(e.scal, self._shard_weights(sh), out=slot)
                sh.comm(e, update)
                e.publish_scalars(slot)
            if adam_table is not None:
                e.adam_from_table(*adam_table)
            else:
                e.adam(pl.segs)
            if part == 'splitc':
                main.wait_stream(e.side_stream)
            return
        if b.N > 0:
            self._forward_backward(w, b.N, vi_, fi_, update, sh=sh, smooth_ok=b.is_full,
                                   extra_losses=(lambda: self._inst_term(sh, update)) if pl.has_inst else None,
                                   publish=pl.early, part=part,
                                   adam_segs=adam_table[0] if adam_table is not None else 0,
                                   use_vposer=self.VERSION >= 1, use_gmm=True)
        elif part != 'tail':        # a shard may own none of a minibatch's samples
            e.scal.zero_()
            if update:
                e.grads.zero_()
            if pl.has_inst:
                self._inst_term(sh, update)
        if adam_table is not None and run_adam:
            e.adam_from_table(*adam_table)
            # (the split-precision chain's weight records are NOT refreshed here for the next forward: measured on one box, the pass over
            #  the weights behind Adam -- serial, 13 - 21 us -- against beside the next step's phase kernel: 1.030 - 1.038 against
            #  1.028 - 1.031 ms per step)

    def _run_part(self, w, b, pl, part):
        """One launch of `part` of the step: a replayed HIP graph once the variant has been seen often enough."""
        e = self.engine
        # (un-captured launches -- instrumented passes, NEMO_GRAPHS=0, counter passes under a profiler -- stage their inputs like
        #  captured ones: the per-view counts that travel with the staged indices select the fused key-point launch, so every
        #  way of running a step launches the same kernels)
        if part in ('all', 'head', 'k0', 'allc', 'splitc', 'bucketc'):          # the first launch of a step stages its inputs
            self._stage_indices(w, b.vi, b.fi, b.N, b.Nv if b.padded else None, full=b.is_full)
        svi, sfi = w['vi_static'][:b.N], w['fi_static'][:b.N]
        if not pl.graphable:
            self._body(w, b, pl, svi, sfi, None, part)
            b.vi, b.fi = svi, sfi
            return
        table = e.adam_table_sync(pl.segs) if (pl.in_graph_adam and part != 'head') else None
        try:
            self._captured(w, self._graph_key(b, pl, part), lambda: self._body(w, b, pl, svi, sfi, table, part),
                           sharded=b.sh.comm is not None, comm_inside=part in ('allc', 'splitc', 'bucketc'))
        except BaseException:
            if table is not None:
                e.adam_table_invalidate()     # step_begin may or may not have advanced the device table
            raise
        if table is not None:
            e.adam_table_commit()
        if part != 'head':
            b.vi, b.fi = svi, sfi

    def step(self, view_idx, frame_idx, update=True, full_batch=False, _shard=None):
        """:3511-3598 (V1/V2), :3796-3909 (V3/V4)."""
        e, a = self.engine, self.args
        if self.VERSION == 0 and a.weight_vp_loss:
            # the reference multiplies weight_vp_loss with the (v2v, kl) TUPLE vposer_loss returns (:3330-3332)
            raise TypeError("NemoV0 cannot run with weight_vp_loss != 0 (can't multiply sequence by non-int in the reference)")
        if self.VERSION >= 3 and update:
            self.training = True
        b = self._prepare_batch(view_idx, frame_idx, full_batch, _shard or ShardInfo())
        sh, N = b.sh, b.N
        w = e._ws(max(N, 1))
        e.scal = w['scal']               # (a replayed graph wrote this workspace's slots)
        e.sync_betas()                   # host-side state a captured graph cannot re-read (checkpoint load, eval)
        pl = self._plan_step(b, update)
        if pl.early or pl.mode in ('split', 'allc', 'splitc', 'bucketc') or (pl.mode == 'buckets' and e.early_readback):
            e.arm_scalars()
        need_adam = update and not (pl.graphable and pl.in_graph_adam)
        e.nvalid = w['vi_static'][w['cap']:].data_ptr() if b.padded else None
        try:
            if pl.mode in ('allc', 'splitc', 'bucketc'):       # one captured launch, collectives inside
                self._run_part(w, b, pl, pl.mode)
                s = e.wait_scalars()
            elif pl.mode == 'buckets':
                s = self._bucketed_update(w, b, pl)
            elif pl.mode == 'split':
                # Two halves -- the loss scalars are final after the first, so their (tiny) all-reduce and the hand-over
                # to the host go to the side stream while the main stream continues with the rest of the backward, the
                # gradient all-reduce and Adam; the host then has the global losses before the step ends and prepares
                # the next launch meanwhile.  (Every rank takes the same route whatever its share of the batch, so the
                # collectives line up.)
                self._run_part(w, b, pl, 'head')
                self._reduce_scalars_on_side_stream(sh)         # weights -> small all-reduce -> publish
                if N > 0:
                    self._run_part(w, b, pl, 'tail')
                sh.comm(e, update)                              # shared-gradient all-reduce (main stream)
                e.adam(pl.segs)
                s = e.wait_scalars()
            else:
                self._run_part(w, b, pl, 'all')
                if pl.early:    # the losses arrive while the backward / Adam launches above are still running
                    if need_adam:
                        e.adam(pl.segs)
                    s = e.wait_scalars()
                else:           # sharded default: ONE collective -- the loss scalars ride in the last 8 floats of the
                    #             shared-gradient all-reduce (_reduce_and_read)
                    s = self._reduce_and_read(sh, update, then=(lambda: e.adam(pl.segs)) if need_adam else None)
        finally:
            e.nvalid = e.view_cnt = None
        vi, fi, Nv = b.vi, b.fi, b.Nv
        if b.padded:
            if sh.comm is None:          # (with a collective the scalars come back weighted: _shard_weights)
                s = np.array(s, dtype=np.float32)
                s[[S_KL, S_GMM, S_3D]] *= np.float32(N / Nv)          # kernel means are over the launch size
            vi, fi = vi[:Nv], fi[:Nv]
        N = Nv                           # (everything below speaks about the real samples)
        f32 = np.float32
        kp = f32(s[S_KP])
        v2v = f32(s[S_V2V]) / f32((sh.n_global or N) * e.NV * 3)
        kl, gmm, l3d = f32(s[S_KL]), f32(s[S_GMM]), f32(s[S_3D])
        loss = kp
        if a.weight_vp_loss and self.VERSION >= 1:
            loss = f32(loss + f32(a.weight_vp_loss) * v2v)
        if a.weight_vp_z_loss and self.VERSION >= 1:
            loss = f32(loss + f32(a.weight_vp_z_loss) * kl)
        loss_dict = {'kp_loss': np.asarray(kp)}
        if self.VERSION >= 3:
            inst = f32(s[S_INST]) if pl.has_inst else 0
            if pl.has_inst:
                loss = f32(loss + f32(a.weight_instance_loss) * inst)
            if getattr(a, 'weight_3d_loss', 0):
                loss = f32(loss + f32(a.weight_3d_loss) * l3d)
            loss_dict['instance_loss'] = np.asarray(inst)
            loss_dict['loss_3d'] = np.asarray(l3d)
        if a.weight_gmm_loss:
            loss = f32(loss + f32(a.weight_gmm_loss) * gmm)
        w_s = float(getattr(a, 'weight_smooth', 0) or 0)
        if w_s and b.is_full:                  # optional term, an extra key only when it is switched on
            loss = f32(loss + f32(w_s) * f32(s[S_SMOOTH]))
            loss_dict['smooth_loss'] = np.asarray(f32(s[S_SMOOTH]))
        if self.VERSION == 0:             # :3325-3340: kp_loss, gmm_loss, total_loss
            loss_dict.update(gmm_loss=np.asarray(gmm), total_loss=np.asarray(loss))
        else:
            loss_dict.update(gmm_loss=np.asarray(gmm), vp_recon_loss=np.asarray(v2v), vp_kl_loss=np.asarray(kl),
                             total_loss=np.asarray(loss))
        # Non-scalar outputs.  Evaluation steps (update=False; what the script dumps with joblib) and training steps whose
        # indices the caller handed over as DEVICE tensors (the unchanged reference script, scripts:291-300) get private
        # copies as in the reference.  Training steps driven with host-side draws (fit.run_fit, draw_batch) discard
        # info_dict (`loss_dict, _ = model.step(...)`): there the tensors are materialised only if accessed -- before
        # the next step() call, which reuses the buffers.
        info_dict = _LazyInfo({'view_idx': vi, 'frame_idx': fi})
        if N > 0:
            makers = dict(loss_all=lambda: self._loss_all(w, N), points2d_gt=lambda: e.targets[vi, fi],
                          points2d=lambda: w['p2d'][:N].clone(), j=lambda: w['j3d'][:N].clone())
            if update and (b.host_idx or b.is_full):
                info_dict.lazy.update(makers)
            else:
                info_dict.update({k: f() for k, f in makers.items()})
        if update and self.schedulers:
            for sch in self.schedulers:
                sch.step(float(loss))
        self.training = False
        return loss_dict, info_dict

    def _bucketed_update(self, w, b, pl):
        """Sharded update step with the shared gradient reduced in THREE buckets, each as soon as the backward has
        completed it (engine.ParamLayout.buckets: heads + layer 4, layer 2, layer 0 + RBF widths + the loss scalars),
        on the engine's communication stream, with the fused Adam of a bucket right behind its collective there.  The
        step is three launches (captured graphs k0 / k1 / k2, cut where a bucket completes); the main stream only joins
        the communication stream at the end of the step.  Every rank issues the same three collectives whatever its
        share of the batch (a rank without samples reduces zeros)."""
        e, sh, segs, N = self.engine, b.sh, pl.segs, b.N
        main, cs = torch.cuda.current_stream(), e.comm_stream
        bk = e.layout.buckets()
        lo, hi = bk[2][0], bk[0][1]
        wv = self._shard_weights(sh)
        slot = e.view('_comm_scalars', e.grads)

        def reduce(i, last=False):
            cs.wait_event(main.record_event())
            with torch.cuda.stream(cs):
                if last:
                    torch.mul(e.scal, wv, out=slot)           # local loss scalars -> this rank's share of the global ones
                sh.comm_bucket(e.grads[bk[i][0]:bk[i][1]])
                if last and e.early_readback:
                    e.publish_scalars(slot)
                e.adam(clip_segments(segs, *bk[i]))

        if N > 0:
            self._run_part(w, b, pl, 'k0')
        else:                    # a shard may own none of a minibatch's samples
            e.scal.zero_()
            e.grads.zero_()
            if pl.has_inst:
                self._inst_term(sh, True)
        reduce(0)
        if N > 0:
            self._run_part(w, b, pl, 'k1')
        reduce(1)
        if N > 0:
            self._run_part(w, b, pl, 'k2')
        reduce(2, last=True)
        # this rank's private parameters (cameras, phase networks, instance codes) on the main stream meanwhile
        e.adam(clip_segments(segs, 0, lo) + clip_segments(segs, hi, e.layout.total))
        main.wait_stream(cs)              # (the next step's forward reads the updated shared parameters)
        if e.early_readback:
            return e.wait_scalars()
        e.scal.copy_(slot)
        return e.read_scalars()

    def _weights_key(self):
        a = self.args
        return (a.loss, float(a.weight_vp_loss), float(a.weight_vp_z_loss), float(a.weight_gmm_loss),
                float(getattr(a, 'weight_3d_loss', 0) or 0), float(getattr(a, 'weight_instance_loss', 0) or 0),
                float(getattr(a, 'weight_smooth', 0) or 0))

    def _shard_weights(self, sh):
        """Per-rank weights that turn the local loss scalars into this rank's share of the global ones (slot 7: the
        NaN-gradient count of a warm-up step -- summed over the ranks as it is)."""
        key = (sh.kr, sh.mr, sh.vr)
        if not hasattr(self, '_shard_w'):
            self._shard_w = {}
        wv = self._shard_w.get(key)
        if wv is None:                       # (cached: a host-to-device upload per step otherwise)
            if len(self._shard_w) > 256:
                self._shard_w.clear()
            wv = self._shard_w[key] = torch.tensor([sh.kr, 1.0, sh.mr, sh.mr, sh.mr, sh.vr, 1.0, 1.0],
                                                   device=self.device)
        return wv

    def _reduce_scalars_on_side_stream(self, sh):
        """Sharded step, after its first half: weight the local loss scalars, all-reduce the 8 floats and
        publish them to the host -- all on the side stream, the main stream goes on with the backward."""
        e = self.engine
        main, side = torch.cuda.current_stream(), e.side_stream
        wv = self._shard_weights(sh)
        if not hasattr(self, '_scal_red'):
            self._scal_red = torch.zeros(8, dtype=torch.float32, device=self.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.mul(e.scal, wv, out=self._scal_red)
            sh.comm_small(self._scal_red)
            e.publish_scalars(self._scal_red)

    def _reduce_and_read(self, sh, update, then=None):
        """Loss scalars to the host (one 32-byte transfer per step).  Sharded: the scalars are weighted
        into global terms, ride along the shared-gradient all-reduce, and are read back reduced.
        ``then`` (the Adam launch) is enqueued before the host waits for the values when they travel
        through pinned memory (engine.publish_scalars), after the synchronisation otherwise."""
        e = self.engine
        slot = None
        if sh.comm is not None:
            wv = self._shard_weights(sh)
            slot = e.view('_comm_scalars', e.grads)
            torch.mul(e.scal, wv, out=slot)
            sh.comm(e, update)
            if not e.early_readback:
                e.scal.copy_(slot)
        if e.early_readback:
            e.arm_scalars()
            e.publish_scalars(slot)          # (reduced scalars straight from the all-reduced buffer: no copy back)
            if then is not None:
                then()
            return e.wait_scalars()
        s = e.read_scalars()
        if then is not None:
            then()
        return s

    def _loss_all(self, w, N):
        Wd = 1 if LOSS_TYPES[self.args.loss] in (2, 3, 5) else 2
        return w['loss_all'].reshape(-1)[:N * self.engine.ctx.n_out * Wd].reshape(N, -1, Wd).clone()

    def draw_batch(self):
        """scripts/learned_multi_view_recon_nn.py:291-296: CPU global RNG, views first then frames."""
        B = self.args.batch_size
        return (torch.randint(0, self.num_views, size=(B,)), torch.randint(0, self.num_frames, size=(B,)))

    # ------------------------------------------------------------------ warm-up (:3455-3509)
    def _warmup_device_work(self, w, N, vi, fi, sh, begin, nan_out=None):
        """One warm-up iteration up to (not including) the update: MLP forward, masked robust 3-D pose loss against the
        HMR / VIBE track, backward to every motion / phase parameter, NaN-gradient count (+= into ``nan_out``, default:
        slot 7 of the step's scalars)."""
        self.engine.batch_sorted = False                        # (random batches: engine.batch_sorted)
        e = self.engine
        st = _stream()
        e.scal = w['scal']
        e.forward_pose(w, N, vi, fi, begin=begin)
        check(e.lib.nemo_pose3d_fwd_bwd(N, 69, w['AA'].data_ptr() + 12, 72, dptr(e.hmr_theta), dptr(e.hmr_mask),
                                        dptr(vi), dptr(fi), e.T, e.scal.data_ptr() + 4 * S_3D, float(sh.mr),
                                        w['dAA'].data_ptr() + 12, 72, None, st), 'nemo_pose3d_fwd_bwd')
        # rot6d backward of the axis-angle gradient; row N (the phase-0 row: a workspace shared with larger batches keeps
        # their row there) cleared in the same launch
        check(e.lib.nemo_pose_bwd_fused(N, dptr(w['ROT']), HEAD_LD, 1, None, dptr(w['dAA']), dptr(w['dROT']), HEAD_LD,
                                        None, None, 0.0, None, HEAD_LD, 1, e.head_meta(w, True), st), 'nemo_pose_bwd_fused')
        e.backward_mlp(w, N, vi, fi, None, has_trans_grad=False)
        check(e.lib.nemo_nan_count(e.grads.data_ptr(), e.grads.numel(),
                                   e.scal.data_ptr() + 4 * S_NAN if nan_out is None else dptr(nan_out), _stream()), 'nemo_nan_count')

    def warmup(self, warmup_steps=1000, _sharder=None):
        """:3455-3509: fit the MLP pose output to the HMR/VIBE 3-D pose (motion + phase optimisers).

        Single process: the iterations never need the host -- their batches are drawn up front in the reference's RNG
        order (views, then frames, per iteration) and live on the device, an iteration is ONE replayed HIP graph (index
        gather by a device counter, zero-fills + Adam-table bookkeeping in the phase kernel's launch, forward, backward,
        NaN-gradient count, loss into a device log, Adam skipped once a NaN has been counted), and the losses and the
        NaN flag are read once at the end.  A NaN gradient raises FloatingPointError as before -- at the end of the
        phase, with the parameters as they were before the first poisoned update."""
        if warmup_steps == 0:
            return []
        e, a = self.engine, self.args
        if self.VERSION == 0:
            # :3207-3269 fits learned_poses to sequences[v]['spin_theta'], which the shipped loader no longer produces
            # (nemo/multi_view_sequence.py:336-392 commented out): the reference raises KeyError there
            raise NotImplementedError("NemoV0.warmup needs 'spin_theta' tracks, which the reference's loader no longer fills")
        if a.batch_size <= -1:
            raise NotImplementedError()
        lm = 'learned_motion.'
        active = set(e.layout.groups['motion'] + e.layout.groups['phase']) - {
            lm + 'linear_out.weight', lm + 'linear_out.bias'}            # their .grad is None in the reference
        opts = [self.opt_motion, self.opt_phase]
        if _sharder is None and self.use_graphs and e.timers is None and a.batch_size > 0:
            return self._warmup_captured(warmup_steps, active, opts)
        losses = []
        for _ in range(warmup_steps):
            if _sharder is not None:          # sharded: GLOBAL draw, routed to this rank's samples
                vi, fi, sh = _sharder()
            else:
                (vi, fi), sh = self.draw_batch(), ShardInfo()
            vi, fi = self._idx(vi), self._idx(fi)
            N = vi.numel()
            w = e._ws(max(N, 1))
            e.scal = w['scal']
            if N > 0:
                self._warmup_device_work(w, N, vi, fi, sh, begin=(w['zero_arena'], True, 0))
            else:
                e.scal.zero_()
                e.grads.zero_()
            s = self._reduce_and_read(sh, True)
            if s[S_NAN] != 0:
                raise FloatingPointError('nan gradient found during warmup')        # :3497-3500
            self._adam_all(opts, active)
            losses.append(float(s[S_3D]))
        return losses

    def _warmup_captured(self, steps, active, opts):
        e, a = self.engine, self.args
        B = a.batch_size
        draws = [self.draw_batch() for _ in range(steps)]              # the reference's RNG order
        seq = e.sequence(steps, B, torch.stack([d[0] for d in draws]), torch.stack([d[1] for d in draws]))
        w = e._ws(B)
        sh = ShardInfo()

        def body(table):
            check(e.lib.nemo_seq_gather(dptr(seq['vi']), dptr(seq['fi']), B, dptr(seq['counter']), dptr(w['vi_static']),
                                        dptr(w['fi_static']), _stream()), 'nemo_seq_gather')
            vi, fi = w['vi_static'][:B], w['fi_static'][:B]
            # (the count is sticky across iterations: once a NaN gradient has been seen no further update is applied)
            self._warmup_device_work(w, B, vi, fi, sh, begin=(w['zero_arena'], True, table[0]), nan_out=seq['nan'])
            check(e.lib.nemo_seq_log(e.scal.data_ptr(), 8, dptr(seq['log']), 8, dptr(seq['counter']), _stream()), 'nemo_seq_log')
            e.adam_from_table(*table, guard=seq['nan'])

        w['_static_src'] = None          # (the gather overwrites the staged indices of a step that used this workspace)
        for _ in range(steps):
            segs = []
            for o in opts:
                segs += o.segments(active)
            table = e.adam_table_sync(segs)
            key = ('warmup', B, seq['gen'], e.ctx.skin_sparse_flag, e.start_global_traj_anywhere, e.detach_articulation,
                   tuple((s_['offset'], s_['numel']) for s_ in segs))
            try:
                self._captured(w, key, lambda: body(table))
            except BaseException:
                e.adam_table_invalidate()
                raise
            e.adam_table_commit()
        torch.cuda.current_stream().synchronize()
        log = seq['log'][:steps].cpu().numpy()
        if float(seq['nan'].cpu()) != 0:
            raise FloatingPointError('nan gradient found during warmup')        # :3497-3500
        return [float(x) for x in log[:, S_3D]]

    # ------------------------------------------------------------------ camera fit (:2869-2906)
    def opt_cam(self, cam_opt_steps=2000, _shard=None):
        """:2869-2906: a fresh Adam on the cameras only, first frame of every view.

        Nothing but the cameras changes between the iterations, and the 3-D side of the objective -- phase / MLP forward,
        FK, the pre-contracted mesh functionals -- does not depend on them: it is evaluated ONCE (the reference recomputes
        the identical values every iteration).  An iteration is then projection + 2-D loss, its camera gradient and the
        fused Adam: one replayed HIP graph that logs its loss on the device; the host reads the log at the end (sharded:
        the per-rank logs are summed with one collective instead of one per iteration)."""
        e, a = self.engine, self.args
        sh = _shard or ShardInfo()
        if cam_opt_steps == 0:
            return []
        m, v = e.scratch_moments()
        cam_opt = FusedAdam(e, ['learned_cameras'], [self.learned_cameras], a.lr_camera, exp_avg=m,
                            exp_avg_sq=v)
        N = self.num_views
        w = e._ws(N)
        if '_cam_idx' not in w:           # (persistent: the captured iteration holds their addresses)
            w['_cam_idx'] = (torch.arange(N, device=self.device), torch.zeros(N, dtype=torch.long, device=self.device))
        vi, fi = w['_cam_idx']
        e.scal = w['scal']
        seq = e.sequence(cam_opt_steps, 0)
        a0, b0 = e.layout.span(['learned_cameras'])
        cam_grads = e.grads[a0:b0]
        # loop-invariant part: pose, FK, mesh functionals of the V first frames
        e.forward_pose(w, N, vi, fi, train=False)
        Mq = e.joint_functionals(w, N)
        lt_weight = float(sh.mr)

        def body(table):
            e.scal = w['scal']
            e.step_begin(w['zero_arena'], cam_grads, table[0] if table is not None else 0)
            e.project_and_loss(w, N, vi, fi, Mq, mean_mode=1)
            e.backward_kp(w, N, vi, fi, Mq, mean_mode=1, upstream=lt_weight, cams_only=True)
            check(e.lib.nemo_seq_log(e.scal.data_ptr(), 8, dptr(seq['log']), 8, dptr(seq['counter']), _stream()), 'nemo_seq_log')
            if table is not None:
                e.adam_from_table(*table, exp_avg=m, exp_avg_sq=v)

        graphable = self.use_graphs and e.timers is None
        for _ in range(cam_opt_steps):
            segs = cam_opt.segments(None)
            if graphable:
                table = e.adam_table_sync(segs)
                try:
                    self._captured(w, ('opt_cam', N, seq['gen'], lt_weight, a.loss, e.start_global_traj_anywhere),
                                   lambda: body(table), sharded=sh.comm is not None)
                except BaseException:
                    e.adam_table_invalidate()
                    raise
                e.adam_table_commit()
            else:
                body(None)
                e.adam(segs, m, v)
        torch.cuda.current_stream().synchronize()
        log = seq['log'][:cam_opt_steps, S_KP].clone()
        if sh.comm is not None:              # this rank's views' share of the per-view mean -> the global loss
            log.mul_(float(sh.mr))
            if sh.comm_log is not None:
                sh.comm_log(log)
        return [np.asarray(x) for x in log.cpu().numpy()]


class NemoV0(MultiViewModel):
    """:3127-3362 (legacy): separate RotNet pose / RotNet orient / FCNN translation networks on the bare warped phase,
    five optimisers, 2-D keypoint + GMM terms (the reference can only run it with weight_vp_loss == 0, and its
    warm-up needs 'spin_theta' tracks the shipped loader no longer produces)."""
    VERSION = 0


class NemoV1(MultiViewModel):
    """:3364-3672: one MLP for pose + orient + trans, raw phase input."""
    VERSION = 1


class NemoV2(NemoV1):
    """:3675-3781: V1 + RBF phase embedding."""
    VERSION = 2

    @property
    def detach_articulation(self):
        return self.engine.detach_articulation

    @detach_articulation.setter
    def detach_articulation(self, v):
        self.engine.detach_articulation = bool(v)

    @property
    def start_global_traj_anywhere(self):
        return self.engine.start_global_traj_anywhere

    @start_global_traj_anywhere.setter
    def start_global_traj_anywhere(self, v):
        self.engine.start_global_traj_anywhere = bool(v)


class NemoV3(NemoV2):
    """:3786-3956: V2 + instance-code regulariser + 3-D pose loss + code noise."""
    VERSION = 3


class NemoV4(NemoV3):
    """:3959-4151: V3 with joints 0..24 and a stochastic camera phase."""
    VERSION = 4

    def opt_cam(self, cam_opt_steps=2000, _sharder=None):
        """:4060-4151: random batches, body pose detached, every optimiser steps.  Single process: one replayed HIP graph
        per iteration (the 2-D / 3-D part of the step without the VPoser and mesh terms, Adam inside), no read-back --
        the reference returns no losses from this phase."""
        e, a = self.engine, self.args
        if a.batch_size <= -1 and cam_opt_steps:
            raise NotImplementedError()
        for _ in range(cam_opt_steps):
            if _sharder is not None:
                vi, fi, sh = _sharder()
            else:
                (vi, fi), sh = self.draw_batch(), ShardInfo()
            b = self._prepare_batch(vi, fi, False, sh)
            N = b.N
            w = e._ws(max(N, 1))
            e.scal = w['scal']
            if N > 0 and sh.comm is None and self.use_graphs and e.timers is None and not b.noise:
                svi, sfi = self._stage_indices(w, b.vi, b.fi, N)
                segs = []
                for o in self.optimizers:
                    segs += o.segments(None)
                table = e.adam_table_sync(segs)

                def body():
                    e.batch_sorted = False                      # (random batches)
                    self._forward_backward(w, N, svi, sfi, True, use_vposer=False, detach_pose=True, sh=sh, adam_segs=table[0])
                    e.adam_from_table(*table)
                key = ('cam4', N, e.detach_articulation, e.start_global_traj_anywhere, self._weights_key(), e.ctx.skin_sparse_flag,
                       e.view_cnt is not None, tuple((s_['offset'], s_['numel']) for s_ in segs))
                try:
                    self._captured(w, key, body)
                except BaseException:
                    e.adam_table_invalidate()
                    raise
                finally:
                    e.view_cnt = None
                e.adam_table_commit()
                continue
            vi, fi = self._idx(b.vi), self._idx(b.fi)
            if N > 0:
                e.batch_sorted = False
                self._forward_backward(w, N, vi, fi, True, use_vposer=False, detach_pose=True, sh=sh)
            else:
                e.scal.zero_()
                e.grads.zero_()
            if sh.comm is not None:
                self._reduce_and_read(sh, True)
            self._adam_all(self.optimizers)
        return []


def make_init_state(args, version, V, img_d0):
    """Initial parameters with the distributions of :3375-3402, :106-126, monotonic_network.py:11-21,
    drawn from the CPU global RNG (bit-identical initial states are obtained by loading a reference
    ``state_dict``).  Returned on the CPU under the reference's state_dict names, so a
    sharded run can build the global state on every rank and keep its slice."""
    C = args.instance_code_size if version >= 1 else 0
    D = args.phase_rbf_dim if version >= 2 else 0
    h, K = args.h_dim, args.monotonic_network_n_nodes
    din = (D if D > 0 else 1) + C
    st = OrderedDict()
    cams = 1e-4 * torch.randn(V, 9)
    cams[:, 3] += 1
    cams[:, 6] += 1
    cams[:, 2] += 2 * FOCAL_LENGTH / (img_d0 * 1 + 1e-9)
    st['learned_cameras'] = cams
    if C > 0:
        st['learned_instance_code'] = 1e-4 * torch.randn(V, C)
    if version == 0:                                    # RotNet(1, h, 23), RotNet(1, h, 1), FCNN(1, h, 3): :3148-3162
        for net, nj in (('learned_poses', 23), ('learned_orient', 1)):
            for name, (fo, fi) in (('net.net.0', (h, 1)), ('net.net.2', (h, h)), ('net.net.4', (h, h)),
                                   ('linear', (nj * 6, h))):
                lin = nn.Linear(fi, fo)
                if name == 'linear':                    # init_last_layer_zero, :86-92
                    nn.init.xavier_uniform_(lin.weight, gain=0.00001)
                    lin.bias.data = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(nj)
                st[f'{net}.{name}.weight'] = lin.weight.data.clone()
                st[f'{net}.{name}.bias'] = lin.bias.data.clone()
        for name, (fo, fi) in (('net.0', (h, 1)), ('net.2', (h, h)), ('net.4', (3, h))):
            lin = nn.Linear(fi, fo)
            st[f'learned_trans.{name}.weight'] = lin.weight.data.clone()
            st[f'learned_trans.{name}.bias'] = lin.bias.data.clone()
    for name, (fo, fi) in (() if version == 0 else (('net.net.0', (h, din)), ('net.net.2', (h, h)), ('net.net.4', (h, h)),
                                                    ('rot_out', (144, h)), ('linear_out', (3, h)))):
        lin = nn.Linear(fi, fo)
        if name == 'rot_out':
            nn.init.xavier_uniform_(lin.weight, gain=0.00001)
            lin.bias.data = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(24)
        st[f'learned_motion.{name}.weight'] = lin.weight.data.clone()
        st[f'learned_motion.{name}.bias'] = lin.bias.data.clone()
    for i in range(V):
        sh = torch.linspace(0, 1, K) if args.phase_init == 'linear' else torch.rand(K)
        st[f'phase_networks.{i}.shifts'] = sh.clamp_(0, 1)
        st[f'phase_networks.{i}.scales'] = torch.full((K,), 15.0)
    if D > 0:
        st['phase_rbf.log_sigmas'] = torch.zeros(D)
    return st


def collate_gt_2d(seqs, label_type='op', thr=30.0):
    """:2908-2961 -> targets (V,T,25,3), bbox diagonal (V,T) + 1e-4."""
    gt = []
    for v in range(seqs.num_views):
        s = seqs.sequences[v]
        if label_type == 'op':
            gt.append(np.array(s['pose_2d_op']))
        elif label_type == 'gt':
            gt.append(np.array(s['pose_2d_gt']))
        elif label_type in ('vibe', 'pare', 'vs'):
            gt.append(np.array(s[label_type + '_joints2d']))
        elif label_type == 'intersection':
            g1, g2 = np.array(s['pose_2d_op']), np.array(s['pose_2d_gt'])
            mean = (g1 + g2)[..., :2] / 2
            dist = np.sqrt(np.power(g1[..., :2] - g2[..., :2], 2).sum(-1, keepdims=True))
            conf = (dist < thr).astype('float32') * g1[..., -1:]
            gt.append(np.concatenate([mean, conf], -1))
        else:
            raise ValueError(label_type)
    pts = torch.tensor(np.array(gt)).float()
    d0 = pts[..., 0].max(-1)[0] - pts[..., 0].min(-1)[0]
    d1 = pts[..., 1].max(-1)[0] - pts[..., 1].min(-1)[0]
    return pts, torch.sqrt(d0 ** 2 + d1 ** 2) + 1e-4


NEMO_VERSIONS = {0: NemoV0, 1: NemoV1, 2: NemoV2, 3: NemoV3, 4: NemoV4}
