"""ORACLE (test infrastructure only) -- model-level CPU restatement of NemoV1..V4.

Restates ``nemo/neural_motion_model.py`` :2758-2843 (losses), :2869-2906 (opt_cam),
:2963-3124 (SMPL call, phases, projection), :3364-3781 (NemoV1/V2), :3786-3956 (NemoV3),
:3959-4151 (NemoV4) as plain PyTorch autograd + ``torch.optim`` on CPU.  Parameters are
kept in a flat dict under the reference's ``state_dict`` names.  Pinned against
``tests/golden/model_*.npz`` (trajectories recorded from the real reference).

Not a product path: see oracle/ops.py header.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops

FOCAL_LENGTH = 5000.0   # hmr/hmr_constants.py:1


def _xavier_uniform(shape, gain, gen=None):
    fan_out, fan_in = shape
    bound = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def _linear_default(fin, fout, gen=None):
    bound = 1.0 / math.sqrt(fin)
    return ((torch.rand(fout, fin, generator=gen) * 2 - 1) * bound,
            (torch.rand(fout, generator=gen) * 2 - 1) * bound)


def collate_gt_2d(seqs, label_type='op', thr=50.0):
    """nemo/neural_motion_model.py:2908-2961."""
    gt = []
    for v in range(seqs.num_views):
        s = seqs.sequences[v]
        if label_type == 'op':
            gt.append(np.array(s['pose_2d_op']))
        elif label_type == 'gt':
            gt.append(np.array(s['pose_2d_gt']))
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


class OracleNemo:
    def __init__(self, version, args, seqs, assets, vposer_sd, gmm, state=None, seed=0):
        assert version in (0, 1, 2, 3, 4)
        self.version, self.args = version, args
        self.V, self.T = seqs.num_views, seqs.num_frames
        self.num_views, self.num_frames = self.V, self.T
        self.IMG_D0, self.IMG_D1 = seqs.IMG_D0, seqs.IMG_D1
        self.smpl = ops.SMPLOracle(assets)
        self.vp = ops.VPoserOracle(vposer_sd)
        self.prior = ops.GMMPriorOracle(gmm)
        self.points2d_gt_all, self.gt_bbox_size = collate_gt_2d(
            seqs, args.label_type, getattr(args, 'label_intersection_threshold', 50.0))
        pose = torch.tensor(np.array([np.array(seqs.sequences[v]['pose']) for v in range(self.V)])).float()
        self.hmr_theta, self.hmr_mask = pose[..., 3:-1], pose[..., -1:]              # :3441-3453
        self.C = args.instance_code_size if version >= 1 else 0       # (NemoV0: no instance code, :3144-3146)
        self.D = args.phase_rbf_dim if version >= 2 else 0
        self.training = False
        self.P = OrderedDict()
        self._init_params(seed)
        if state is not None:
            self.load_state(state)
        self._build_optimizers()

    # ------------------------------------------------------------------ parameters
    def _init_params(self, seed):
        """Same distributions as :3375-3402 / :120-126 / monotonic_network.py:11-21 (not the
        same RNG stream -- parity tests load the reference's recorded initial state)."""
        a, g = self.args, torch.Generator().manual_seed(seed)
        cams = 1e-4 * torch.randn(self.V, 9, generator=g)
        cams[:, 3] += 1
        cams[:, 6] += 1
        cams[:, 2] += 2 * FOCAL_LENGTH / (self.IMG_D0 * 1 + 1e-9)
        self.P['learned_cameras'] = cams
        if self.C > 0:
            self.P['learned_instance_code'] = 1e-4 * torch.randn(self.V, self.C, generator=g)
        din = (self.D if self.D > 0 else 1) + self.C
        h = a.h_dim
        if self.version == 0:
            # :3148-3162: RotNet(1, h, 23) poses, RotNet(1, h, 1) orient (FCNN + Linear, last layer xavier(1e-5) and an
            # identity-6d bias, :74-94), FCNN(1, h, 3) translation
            for net, nj in (('learned_poses', 23), ('learned_orient', 1)):
                for name, (fi, fo) in (('net.net.0', (1, h)), ('net.net.2', (h, h)), ('net.net.4', (h, h))):
                    w, b = _linear_default(fi, fo, g)
                    self.P[f'{net}.{name}.weight'], self.P[f'{net}.{name}.bias'] = w, b
                self.P[f'{net}.linear.weight'] = _xavier_uniform((nj * 6, h), 1e-5, g)
                self.P[f'{net}.linear.bias'] = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(nj)
            for name, (fi, fo) in (('net.0', (1, h)), ('net.2', (h, h)), ('net.4', (h, 3))):
                w, b = _linear_default(fi, fo, g)
                self.P[f'learned_trans.{name}.weight'], self.P[f'learned_trans.{name}.bias'] = w, b
        else:
            for name, (fi, fo) in (('net.net.0', (din, h)), ('net.net.2', (h, h)), ('net.net.4', (h, h)),
                                   ('rot_out', (h, 144)), ('linear_out', (h, 3))):
                w, b = _linear_default(fi, fo, g)
                self.P[f'learned_motion.{name}.weight'] = w
                self.P[f'learned_motion.{name}.bias'] = b
            self.P['learned_motion.rot_out.weight'] = _xavier_uniform((144, h), 1e-5, g)
            self.P['learned_motion.rot_out.bias'] = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(24)
        self.P['learned_betas'] = torch.zeros(1, 10)
        K = a.monotonic_network_n_nodes
        for i in range(self.V):
            sh = torch.linspace(0, 1, K) if a.phase_init == 'linear' else torch.rand(K, generator=g)
            self.P[f'phase_networks.{i}.shifts'] = sh.clamp(0, 1)
            self.P[f'phase_networks.{i}.scales'] = torch.ones(K) * 15
        if self.D > 0:
            self.P['phase_rbf.log_sigmas'] = torch.zeros(self.D)
            self.rbf_centres = torch.linspace(0, 1, self.D).unsqueeze(1)
        for k in self.P:
            self.P[k] = self.P[k].clone().requires_grad_(True)

    def to(self, device):
        """Moves every tensor of the oracle to `device` and rebuilds the optimisers (fresh Adam state).
        Used by bench.py's `torch_gpu_baseline` leg: the same unfused PyTorch restatement of the reference
        step, run through PyTorch-ROCm's stock kernels on the GPU the HIP engine is measured on.  Callers
        wrap `step()` in `with torch.device(device):` so that the small tensors it creates land there."""
        mv = lambda t: t.to(device) if torch.is_tensor(t) else t
        self.P = OrderedDict((k, v.detach().to(device).requires_grad_(True)) for k, v in self.P.items())
        if self.D > 0:
            self.rbf_centres = mv(self.rbf_centres)
        self.smpl.a = {k: mv(v) for k, v in self.smpl.a.items()}
        self.vp.sd = {k: mv(v) for k, v in self.vp.sd.items()}
        for k in ('means', 'precisions', 'nll_weights'):
            setattr(self.prior, k, mv(getattr(self.prior, k)))
        for k in ('points2d_gt_all', 'gt_bbox_size', 'hmr_theta', 'hmr_mask'):
            setattr(self, k, mv(getattr(self, k)))
        self._build_optimizers()
        return self

    def load_state(self, state):
        for k, v in state.items():
            if k in self.P:
                with torch.no_grad():
                    self.P[k].copy_(torch.as_tensor(v))

    def state_dict(self):
        sd = OrderedDict((k, v.detach().clone()) for k, v in self.P.items())
        if self.D > 0:
            sd['phase_rbf.centres'] = self.rbf_centres.clone()
        return sd

    def _build_optimizers(self):
        a, P = self.args, self.P
        if self.version == 0:                                                          # :3171-3206
            cls = torch.optim.Adam if a.opt_human == 'adam' else torch.optim.AdamW
            grp = lambda pre: [P[k] for k in P if k.startswith(pre)]
            self.opt_cameras = torch.optim.Adam([P['learned_cameras']], lr=a.lr_camera, weight_decay=0)
            self.opt_poses = cls(grp('learned_poses.'), lr=a.lr_pose, weight_decay=a.wd_human)
            self.opt_orient = cls(grp('learned_orient.'), lr=a.lr_orient, weight_decay=a.wd_human)
            self.opt_trans = torch.optim.Adam(grp('learned_trans.'), lr=a.lr_trans, weight_decay=0.0)
            self.opt_phase = torch.optim.Adam(grp('phase_networks.'), lr=a.lr_phase, weight_decay=0.0)
            self.optimizers = [self.opt_cameras, self.opt_poses, self.opt_orient, self.opt_trans, self.opt_phase]
            self.schedulers = []
            if a.lr_factor < 1:
                self.schedulers = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, factor=a.lr_factor, min_lr=1e-6)
                                   for o in self.optimizers]
            return
        motion = [P[k] for k in P if k.startswith('learned_motion.')]
        if self.D > 0:
            motion.append(P['phase_rbf.log_sigmas'])                                   # :3706-3711
        cls = torch.optim.Adam if a.opt_human == 'adam' else torch.optim.AdamW
        self.opt_cameras = torch.optim.Adam([P['learned_cameras']], lr=a.lr_camera, weight_decay=0)
        self.opt_motion = cls(motion, lr=a.lr_human, weight_decay=a.wd_human)
        self.opt_phase = torch.optim.Adam([P[k] for k in P if k.startswith('phase_networks.')],
                                          lr=a.lr_phase, weight_decay=0.0)
        self.optimizers = [self.opt_cameras, self.opt_motion, self.opt_phase]
        if self.C > 0:
            self.opt_instance = torch.optim.Adam([P['learned_instance_code']], lr=a.lr_instance,
                                                 weight_decay=0)
            self.optimizers.append(self.opt_instance)
        self.schedulers = []
        if a.lr_factor < 1:                                                            # :3433-3438
            self.schedulers = [torch.optim.lr_scheduler.ReduceLROnPlateau(
                o, factor=a.lr_factor, min_lr=1e-6) for o in self.optimizers]

    # ------------------------------------------------------------------ forward pieces
    def _input_phases(self, view_idx, frame_idx, phases=None):
        """:3647-3657.  Only the owning network is evaluated per sample (bit-identical to
        evaluating all V and gathering)."""
        if phases is None:
            raw = torch.linspace(0, 1, self.T)[frame_idx].unsqueeze(1)
        else:
            raw = phases.unsqueeze(1)
        sh = torch.stack([self.P[f'phase_networks.{i}.shifts'] for i in range(self.V)])[view_idx]
        sc = torch.stack([self.P[f'phase_networks.{i}.scales'] for i in range(self.V)])[view_idx]
        return ops.monotonic_forward(sh, sc, raw)

    def _embed(self, phases):
        if self.D > 0:
            return ops.rbf_forward(self.P['phase_rbf.log_sigmas'], self.rbf_centres, phases,
                                   self.args.rbf_kernel)
        return phases

    def _v0_nets(self, x):
        """NemoV0 (:3005-3034): three networks on the warped phase -- RotNet poses (23 x 6), RotNet orient (6), FCNN
        translation.  Returns rot6d (N, 144) = [orient | poses] and trans (N, 3) like the merged MotionNet."""
        import torch.nn.functional as F
        P = self.P

        def fcnn(pre, x_):
            h_ = F.relu(F.linear(x_, P[pre + '0.weight'], P[pre + '0.bias']))
            h_ = F.relu(F.linear(h_, P[pre + '2.weight'], P[pre + '2.bias']))
            return F.linear(h_, P[pre + '4.weight'], P[pre + '4.bias'])
        rot = lambda net: F.linear(F.relu(fcnn(net + '.net.net.', x)), P[net + '.linear.weight'], P[net + '.linear.bias'])
        return torch.cat([rot('learned_orient'), rot('learned_poses')], 1), fcnn('learned_trans.net.', x)

    def _smpl(self, body_rotmats, orient6d):
        """:2963-2976 with pose_type='rotmat'."""
        R0 = ops.rot6d_to_rotmat(orient6d).unsqueeze(1)
        return self.smpl.forward(self.P['learned_betas'], torch.cat([R0, body_rotmats], 1))

    def get_preds_batch(self, view_idx, frame_idx, add_trans=True, phases=None, detach_pose=False):
        """:3637-3672 / :3733-3781 / :3911-3956 / :3968-4058."""
        N = len(view_idx)
        x = self._embed(self._input_phases(view_idx, frame_idx, phases))
        if self.C > 0:
            codes = self.P['learned_instance_code'][view_idx]
            if self.version >= 3 and self.training and self.args.code_noise > 0:
                codes = codes + self.args.code_noise * torch.randn_like(codes)
            x = torch.cat([x, codes], 1)
        if self.version == 0:
            rot6d, trans = self._v0_nets(x)
        else:
            rot6d, trans = ops.motionnet_forward(self.P, 'learned_motion.', x)
        rotmat = ops.rot6d_to_rotmat(rot6d).view(N, 24, 3, 3)
        pose_aa = ops.rotmat_to_aa(rotmat.reshape(-1, 3, 3)).reshape(N, 72)
        body = rotmat[:, 1:].detach() if detach_pose else rotmat[:, 1:]
        verts, j49, _ = self._smpl(body, rot6d[:, :6])
        x0 = torch.zeros(1, 1)
        x0 = torch.cat([self._embed(x0), torch.zeros(1, self.C)], 1) if self.C > 0 else self._embed(x0)
        trans0 = self._v0_nets(x0)[1] if self.version == 0 else ops.motionnet_forward(self.P, 'learned_motion.', x0)[1]
        trans = trans - trans0
        if add_trans:
            verts, j49 = verts + trans.unsqueeze(1), j49 + trans.unsqueeze(1)
        idx = list(range(0, 25)) if self.version == 4 else [38] + list(range(1, 25))
        return {'view_idx': view_idx, 'frame_idx': frame_idx, 'v': verts, 'j': j49[:, idx],
                'poses': pose_aa[:, 3:], 'orient': rot6d[:, :6], 'orient_aa': pose_aa[:, :3],
                'trans': trans}

    def full_indices(self):
        v = torch.arange(self.V).repeat_interleave(self.T)
        f = torch.arange(self.T).repeat(self.V)
        return v, f

    def learned_camera_projection(self, points3d, view_idx):
        """:3073-3124 (vectorised over views; centre = (IMG_D0//2, IMG_D1//2), :3104-3106)."""
        cams = self.P['learned_cameras'][view_idx]
        R = ops.rot6d_to_rotmat(cams[:, 3:])
        c = torch.tensor([[float(self.IMG_D0 // 2), float(self.IMG_D1 // 2)]]).expand(len(view_idx), -1)
        return ops.perspective_projection(points3d, R, cams[:, :3], FOCAL_LENGTH, c)

    def vposer_loss(self, poses, orient):
        """:2775-2804."""
        N = poses.shape[0]
        mean, scale = self.vp.encode(poses[:, :63])
        dec_aa, _ = self.vp.decode(mean)
        recon = torch.cat([dec_aa.reshape(N, -1), poses[:, 63:]], 1)
        R_orig = ops.batch_rodrigues(poses.reshape(-1, 3)).reshape(N, 23, 3, 3)
        R_rec = ops.batch_rodrigues(recon.reshape(-1, 3)).reshape(N, 23, 3, 3)
        v_orig = self._smpl(R_orig, orient)[0]
        v_rec = self._smpl(R_rec, orient)[0]
        v2v = (v_rec.detach() - v_orig).abs().mean()
        return v2v, ops.kl_to_std_normal(mean, scale)

    def kp_loss(self, pred, view_idx, frame_idx):
        gt = self.points2d_gt_all[view_idx, frame_idx]
        size = self.gt_bbox_size[view_idx, frame_idx]
        loss_all = ops.keypoint_loss(pred, gt[..., :2], gt[..., 2:], size, self.args.loss)
        return loss_all, gt

    # ------------------------------------------------------------------ optimisation API
    def step(self, view_idx, frame_idx, update=True, full_batch=False, shard=None):
        """:3511-3598 (V1/V2) and :3796-3909 (V3/V4).  ``kp_loss`` is the pure (GPU-semantics)
        value, SURVEY.md section 7 quirk (v).

        ``shard`` (tests of the instance-sharding maths only): dict(kr, mr, vr, comm) -- this
        process holds a block of the views; its local terms are scaled to global ones and
        ``comm(shared_grads, scalars)`` sums the shared gradients / loss scalars over the ranks."""
        a = self.args
        kr, mr, vr, comm = 1.0, 1.0, 1.0, None
        if shard is not None:
            kr, mr, vr, comm = shard['kr'], shard['mr'], shard['vr'], shard['comm']
        if self.version >= 3 and update:
            self.training = True
        is_full = not (a.batch_size > -1 and not full_batch)
        if is_full:
            view_idx, frame_idx = self.full_indices()
        N = len(view_idx)
        smooth = None
        zero = torch.zeros(())
        terms = dict(kp=zero, v2v=zero, kl=zero, gmm=zero, l3=zero, inst=zero)
        info = {'view_idx': view_idx, 'frame_idx': frame_idx}
        if N > 0:
            pd = self.get_preds_batch(view_idx, frame_idx)
            p2d = self.learned_camera_projection(pd['j'], view_idx)
            loss_all, gt = self.kp_loss(p2d, view_idx, frame_idx)
            terms['kp'] = kr * ops.per_view_mean_loss(loss_all, gt[..., -1:], view_idx)
            if self.version >= 1:       # (NemoV0 calls vposer_loss too, :3330, but can only run with weight_vp_loss == 0)
                v2v, kl = self.vposer_loss(pd['poses'], pd['orient'])
                terms['v2v'], terms['kl'] = mr * v2v, mr * kl
            if self.version >= 3 and a.weight_3d_loss:
                terms['l3'] = mr * ops.keypoint_loss(pd['poses'], self.hmr_theta[view_idx, frame_idx],
                                                     self.hmr_mask[view_idx, frame_idx], None,
                                                     'mse_robust').mean()
            terms['gmm'] = mr * self.prior(pd['poses']).mean()                          # :2758-2773
            if is_full and getattr(a, 'weight_smooth', 0):
                # OPTIONAL extension (not in the published NemoV* step): HuMoR's joints3d_smooth_loss,
                # humor/humor/fitting/fitting_loss.py:366-370, on the 25 output joints of complete sequences
                jj = pd['j'].reshape(self.V, self.T, -1, 3)
                smooth = ops.joints3d_smooth_loss(jj)
            info.update(loss_all=loss_all.detach(), points2d_gt=gt, points2d=p2d.detach(),
                        j=pd['j'].detach())
        if self.version >= 3 and a.weight_instance_loss:
            terms['inst'] = vr * (self.P['learned_instance_code'] ** 2).mean()
        loss = terms['kp']
        if self.version == 0 and a.weight_vp_loss:
            raise TypeError('NemoV0 multiplies weight_vp_loss with the (v2v, kl) tuple of vposer_loss (:3330-3332)')
        if a.weight_vp_loss and self.version >= 1:
            loss = loss + a.weight_vp_loss * terms['v2v']
        if a.weight_vp_z_loss and self.version >= 1:
            loss = loss + a.weight_vp_z_loss * terms['kl']
        if self.version >= 3:
            if a.weight_instance_loss:
                loss = loss + a.weight_instance_loss * terms['inst']
            if a.weight_3d_loss:
                loss = loss + a.weight_3d_loss * terms['l3']
        if a.weight_gmm_loss:
            loss = loss + a.weight_gmm_loss * terms['gmm']
        if smooth is not None:
            loss = loss + a.weight_smooth * smooth
        if update:
            for o in self.optimizers:
                o.zero_grad()
            if loss.requires_grad:
                loss.backward()
        scal = torch.stack([terms[k].detach().float() for k in ('kp', 'v2v', 'kl', 'gmm', 'l3', 'inst')]
                           + [loss.detach().float()])
        if comm is not None:
            shared = [p for p in self.opt_motion.param_groups[0]['params']]
            for p in shared:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            comm([p.grad for p in shared] if update else [], scal)
        if update:
            for o in self.optimizers:
                for p in o.param_groups[0]['params']:
                    if p.grad is None and shard is not None:
                        p.grad = torch.zeros_like(p)        # absent views still take an Adam step
                o.step()
            for s_ in self.schedulers:
                s_.step(float(scal[6]))
        out = {'kp_loss': scal[0], 'gmm_loss': scal[3], 'vp_recon_loss': scal[1], 'vp_kl_loss': scal[2],
               'total_loss': scal[6]}
        if self.version == 0:                                                          # :3325-3340
            out = {'kp_loss': scal[0], 'gmm_loss': scal[3], 'total_loss': scal[6]}
        if self.version >= 3:
            out['instance_loss'] = scal[5] if a.weight_instance_loss else 0
            if a.weight_3d_loss:
                out['loss_3d'] = scal[4]
        if smooth is not None:
            out['smooth_loss'] = smooth.detach().float()
        loss_dict = {k: np.asarray(v.cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
                     for k, v in out.items()}
        self.training = False
        return loss_dict, info

    def draw_batch(self):
        """scripts/learned_multi_view_recon_nn.py:291-296: views first, then frames, CPU RNG."""
        B = self.args.batch_size
        return torch.randint(0, self.V, size=(B,)), torch.randint(0, self.T, size=(B,))

    def warmup(self, steps):
        """:3455-3509."""
        if self.args.batch_size <= -1 and steps:
            raise NotImplementedError()
        losses = []
        for _ in range(steps):
            vi, fi = self.draw_batch()
            pd = self.get_preds_batch(vi, fi)
            loss = ops.keypoint_loss(pd['poses'], self.hmr_theta[vi, fi], self.hmr_mask[vi, fi],
                                     None, 'mse_robust').mean()
            self.opt_motion.zero_grad()
            self.opt_phase.zero_grad()
            loss.backward()
            self.opt_motion.step()
            self.opt_phase.step()
            losses.append(float(loss.detach()))
        return losses

    def opt_cam(self, steps):
        if self.version == 4:
            return self._opt_cam_v4(steps)
        cam_opt = torch.optim.Adam([self.P['learned_cameras']], lr=self.args.lr_camera)   # :2870
        log = []
        vi = torch.arange(self.V)
        fi = torch.zeros(self.V, dtype=torch.long)
        for _ in range(steps):
            cam_opt.zero_grad()
            pd = self.get_preds_batch(vi, fi)
            p2d = self.learned_camera_projection(pd['j'], vi)
            loss_all, _ = self.kp_loss(p2d, vi, fi)
            loss = loss_all.mean()                                                        # :2865
            loss.backward()
            log.append(float(loss.detach()))
            cam_opt.step()
        return log

    def _opt_cam_v4(self, steps):
        """:4060-4151: stochastic batches, body pose detached, every optimiser steps."""
        a = self.args
        if a.batch_size <= -1 and steps:
            raise NotImplementedError()
        for _ in range(steps):
            vi, fi = self.draw_batch()
            pd = self.get_preds_batch(vi, fi, detach_pose=True)
            p2d = self.learned_camera_projection(pd['j'], vi)
            loss_all, gt = self.kp_loss(p2d, vi, fi)
            loss = ops.per_view_mean_loss(loss_all, gt[..., -1:], vi)
            if a.weight_3d_loss:
                l3 = ops.keypoint_loss(pd['poses'], self.hmr_theta[vi, fi], self.hmr_mask[vi, fi],
                                       None, 'mse_robust').mean()
                loss = loss + a.weight_3d_loss * l3
            for o in self.optimizers:
                o.zero_grad()
            loss.backward()
            for o in self.optimizers:
                o.step()
        return []
