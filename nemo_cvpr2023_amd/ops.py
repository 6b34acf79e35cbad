"""The reference's operator-level call signatures over the C ABI of ``libnemo_hip.so``.

For someone who swaps single operators inside the reference's own code instead of the whole model class: the same names,
argument meaning and return shapes as

    hmr/geometry.py:9-45      batch_rodrigues(theta)
    hmr/geometry.py:47-61     rot6d_to_rotmat(x)
    hmr/geometry.py:78-106    perspective_projection(points, rotation, translation, focal_length, camera_center)
    hmr/geometry.py:181-210   rotation_matrix_to_angle_axis(rotation_matrix)
    hmr/smpl.py:17-43         SMPL(...).forward(betas, body_pose, global_orient, pose2rot) -> .vertices / .joints (49)
    hmr/smplify/prior.py:100-196   MaxMixturePrior(...)(pose, betas) -> per-sample min log-likelihood
    human_body_prior/models/vposer_model.py:89-106   VPoser.encode(pose_body) / .decode(Zin)
    nemo/utils/misc_utils.py:91-105   GMoF(rho)(residual, sqrt)

Every function launches the hand-written HIP kernels (there is no PyTorch fallback: without the extension or a GPU they
raise ``NemoHipError``).  ``rot6d_to_rotmat`` and ``batch_rodrigues`` are ``torch.autograd.Function``s with the kernels'
hand-derived adjoints; ``MaxMixturePrior`` back-propagates through the selected mixture component; the others are
forward-only (``torch.no_grad`` semantics: the fit itself uses the fused kernels of ``engine.FitEngine``, which carry their
own backward).  Model data (SMPL arrays, VPoser weights, GMM) is passed in as arrays -- synthetic in the tests, the
licensed files via ``assets.load_real_assets`` in production.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, dptr


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    if not torch.cuda.is_available():
        raise _lib.NemoHipError('nemo_cvpr2023_amd.ops needs an MI355X (no CPU fallback by design)')
    t = torch.as_tensor(t)
    dev = t.device if t.is_cuda else torch.device('cuda', torch.cuda.current_device())
    return t.detach().to(dev, torch.float32).contiguous()


# ------------------------------------------------------------------------------------------ rotations
class _Rot6dToRotmat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L = _lib.load()
        x6 = _dev(x).reshape(-1, 6)
        R = torch.empty(x6.shape[0], 3, 3, device=x6.device)
        check(L.nemo_rot6d_fwd(x6.shape[0], 1, dptr(x6), 6, 1, dptr(R), None, _st()), 'nemo_rot6d_fwd')
        ctx.save_for_backward(x6)
        ctx.in_shape = x.shape
        return R

    @staticmethod
    def backward(ctx, dR):
        (x6,) = ctx.saved_tensors
        d = torch.empty_like(x6)
        g = dR.contiguous().float()
        check(_lib.load().nemo_rot6d_bwd(x6.shape[0], 1, dptr(x6), 6, 1, dptr(g), None, dptr(d), 6, _st()), 'nemo_rot6d_bwd')
        return d.reshape(ctx.in_shape)


def rot6d_to_rotmat(x):
    """hmr/geometry.py:47-61: (B, 6) [or anything viewable as (-1, 3, 2)] -> (B, 3, 3), Gram-Schmidt on the two columns."""
    return _Rot6dToRotmat.apply(x)


class _BatchRodrigues(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta):
        th = _dev(theta).reshape(-1, 3)
        R = torch.empty(th.shape[0], 3, 3, device=th.device)
        check(_lib.load().nemo_rodrigues_fwd(th.shape[0], dptr(th), 0, dptr(R), _st()), 'nemo_rodrigues_fwd')
        ctx.save_for_backward(th)
        return R

    @staticmethod
    def backward(ctx, dR):
        (th,) = ctx.saved_tensors
        d = torch.empty_like(th)
        g = dR.contiguous().float()
        check(_lib.load().nemo_rodrigues_bwd(th.shape[0], dptr(th), dptr(g), dptr(d), _st()), 'nemo_rodrigues_bwd')
        return d


def batch_rodrigues(theta):
    """hmr/geometry.py:9-45: axis-angle (N, 3) -> rotation matrices (N, 3, 3) through the quaternion, angle = |theta + 1e-8|."""
    return _BatchRodrigues.apply(theta)


@torch.no_grad()
def rotation_matrix_to_angle_axis(rotation_matrix):
    """hmr/geometry.py:181-210: (N, 3, 3) (or (N, 3, 4): the last column is ignored, as in the reference) -> (N, 3); the four
    quaternion branches of :246-346, NaN -> 0."""
    R = _dev(rotation_matrix)
    if R.shape[-1] == 4:
        R = R[..., :3].contiguous()
    R = R.reshape(-1, 3, 3)
    aa = torch.empty(R.shape[0], 3, device=R.device)
    check(_lib.load().nemo_rotmat_to_aa(R.shape[0], dptr(R), 1, dptr(aa), _st()), 'nemo_rotmat_to_aa')
    return aa


# ------------------------------------------------------------------------------------------ projection
@torch.no_grad()
def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """hmr/geometry.py:78-106: points (bs, N, 3), rotation (bs, 3, 3), translation (bs, 3), focal_length (bs,) or scalar,
    camera_center (bs, 2) -> (bs, N, 2).  Runs ``nemo_project`` with one camera per batch entry; its cameras are
    [t | first two columns of R] (rot6d of an orthonormal matrix reproduces the matrix), its intrinsics scalars, so batch
    entries are grouped by (focal, centre)."""
    L = _lib.load()
    pts, R, t = _dev(points), _dev(rotation), _dev(translation)
    bs, n = pts.shape[0], pts.shape[1]
    f = torch.as_tensor(focal_length, dtype=torch.float32).reshape(-1).cpu()
    f = f.expand(bs) if f.numel() == 1 else f
    c = torch.as_tensor(camera_center, dtype=torch.float32).cpu().reshape(-1, 2)
    c = c.expand(bs, 2) if c.shape[0] == 1 else c
    cams = torch.cat([t, R[:, :, :2].reshape(bs, 6)], 1).contiguous()          # rot6d layout: (3, 2) row-major
    out = torch.empty(bs, n, 2, device=pts.device)
    key = torch.cat([f[:, None], c], 1)
    uniq, inv = torch.unique(key, dim=0, return_inverse=True)
    for u in range(uniq.shape[0]):
        rows = torch.nonzero(inv == u).reshape(-1).to(pts.device)
        p_u, cam_u = pts[rows].contiguous(), cams[rows].contiguous()
        vi = torch.arange(rows.numel(), device=pts.device)
        o_u = torch.empty(rows.numel(), n, 2, device=pts.device)
        check(L.nemo_project(rows.numel(), n, rows.numel(), dptr(p_u), dptr(vi), dptr(cam_u), float(uniq[u, 0]),
                             float(uniq[u, 1]), float(uniq[u, 2]), dptr(o_u), _st()), 'nemo_project')
        out[rows] = o_u
    return out


class GMoF(nn.Module):
    """nemo/utils/misc_utils.py:91-105 (elementwise; inside the fit it is part of the keypoint kernels)."""

    def __init__(self, rho=100):
        super().__init__()
        self.rho = rho

    def forward(self, residual, sqrt):
        sq = residual ** 2
        if sqrt:
            sq = torch.sqrt(sq.sum(-1)).unsqueeze(-1)
        return self.rho ** 2 * torch.div(sq, sq + self.rho ** 2)


# ------------------------------------------------------------------------------------------ SMPL
class SMPL(nn.Module):
    """hmr/smpl.py:17-43 over smplx 0.1.28 semantics: ``forward(betas, body_pose, global_orient, pose2rot)`` ->
    ``.vertices (B, NV, 3)``, ``.joints (B, 49, 3)`` (24 FK joints + 21 selector vertices + 9 extra-regressor rows, indexed by
    ``joint_map``).  ``assets``: the arrays smplx / hmr register as buffers (``synthetic.make_smpl_assets`` or the real
    model through ``assets.load_real_assets``)."""

    def __init__(self, assets, device='cuda:0'):
        super().__init__()
        from .engine import SmplContext
        self.device = torch.device(device)
        jm = [int(x) for x in assets['joint_map']]
        # the keypoint kernel serves up to 32 output joints per context: two contexts cover the 49-joint map
        self._ctx = [SmplContext(assets, jm[:25], self.device), SmplContext(assets, jm[25:], self.device)]
        self.NV = self._ctx[0].NV
        self.joint_map = torch.tensor(jm, dtype=torch.long)
        self.faces = np.asarray(assets['faces']) if 'faces' in assets else np.zeros((0, 3), dtype=np.int64)

    @torch.no_grad()
    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kwargs):
        L = _lib.load()
        if body_pose is None:
            raise ValueError('body_pose is required')
        bp = _dev(body_pose)
        B = bp.shape[0]
        if pose2rot:                                         # axis-angle in: lbs.py:303-334 (matrix-form Rodrigues)
            go = torch.zeros(B, 3, device=bp.device) if global_orient is None else _dev(global_orient).reshape(B, 3)
            theta = torch.cat([go, bp.reshape(B, 69)], 1).contiguous()
            R = torch.empty(B, 24, 9, device=bp.device)
            check(L.nemo_rodrigues_fwd(B * 24, dptr(theta), 1, dptr(R), _st()), 'nemo_rodrigues_fwd')
        else:
            go = torch.eye(3, device=bp.device).expand(B, 1, 3, 3) if global_orient is None else \
                _dev(global_orient).reshape(B, 1, 3, 3)
            R = torch.cat([go, bp.reshape(B, 23, 3, 3)], 1).reshape(B, 24, 9).contiguous()
        b = np.zeros(10, dtype=np.float32) if betas is None else np.asarray(torch.as_tensor(betas).detach().cpu().float()).reshape(-1)[:10]
        A, Jp, PF = (torch.empty(B, *s, device=bp.device) for s in ((24, 12), (24, 3), (208,)))
        verts = torch.empty(B, self.NV, 3, device=bp.device)
        joints = []
        one = torch.zeros(1, 9, device=bp.device)
        one[0, 3] = one[0, 6] = 1.0
        one[0, 2] = 10.0
        vi = torch.zeros(B, dtype=torch.long, device=bp.device)
        ws = torch.zeros(16 << 20, device=bp.device)
        for k, ctx in enumerate(self._ctx):
            ctx.set_betas(b)
            check(L.nemo_fk_fwd(ctx.handle, B, dptr(R), dptr(A), dptr(Jp), dptr(PF), 208, _st()), 'nemo_fk_fwd')
            nq72 = max(ctx.nq * 72, 1)
            Mq = torch.zeros(B, nq72, device=bp.device)
            if ctx.nq:
                check(L.nemo_gemm_f32(0, 0, B, ctx.nq * 72, 207, dptr(PF), 208, ctx.C1, ctx.nq * 72, dptr(Mq), nq72, ctx.c0, 0,
                                      None, 0, 0, 1.0, 0, 0, dptr(ws), ws.numel() * 4, _st()), 'nemo_gemm_f32')
            j = torch.empty(B, ctx.n_out, 3, device=bp.device)
            check(L.nemo_kp_fwd(ctx.handle, B, 1, 1, dptr(A), dptr(Jp), dptr(Mq), nq72, None, 3, 0, dptr(vi), None, dptr(one),
                                None, None, 5000.0, 0.0, 0.0, 0, 0, dptr(j), None, None, None, None, _st()), 'nemo_kp_fwd')
            joints.append(j)
            if k == 0:
                ldP = ctx.ldP
                chunk = 2048
                VP = torch.empty(min(B, chunk), ldP, device=bp.device)
                for c0 in range(0, B, chunk):
                    m = min(chunk, B - c0)
                    check(L.nemo_gemm_f32(0, 0, m, 3 * self.NV, 207, PF.data_ptr() + 4 * c0 * 208, 208, ctx.posedirs, ldP,
                                          dptr(VP), ldP, ctx.v_shaped, 0, None, 0, 0, 1.0, 0, 0, dptr(ws), ws.numel() * 4,
                                          _st()), 'nemo_gemm_f32')
                    check(L.nemo_skin_vertices(ctx.handle, m, dptr(VP), ldP, A.data_ptr() + 4 * c0 * 288, None, 3,
                                               verts.data_ptr() + 4 * c0 * 3 * self.NV, _st()), 'nemo_skin_vertices')
        return SimpleNamespace(vertices=verts, joints=torch.cat(joints, 1), betas=betas, body_pose=body_pose,
                               global_orient=global_orient)


# ------------------------------------------------------------------------------------------ priors
class _GMMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, prior):
        L = _lib.load()
        x = _dev(pose).reshape(-1, 69)
        N, c = x.shape[0], prior._c
        per = torch.empty(N, device=x.device)
        dx = torch.zeros(N, 69, device=x.device)
        scal = torch.zeros(1, device=x.device)
        ws = torch.zeros(N, c['M'], device=x.device)
        # scale = N: the kernel's gradient is d(mean over N)/dx, i.e. d ll_s / dx_s / N
        check(L.nemo_gmm_fwd_bwd(N, c['M'], 69, dptr(x), 69, dptr(c['means']), dptr(c['prec']), dptr(c['log_nllw']), dptr(ws),
                                 dptr(scal), dptr(per), float(N), dptr(dx), 69, None, _st()), 'nemo_gmm_fwd_bwd')
        ctx.save_for_backward(dx)
        return per

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g.reshape(-1, 1), None


class MaxMixturePrior(nn.Module):
    """hmr/smplify/prior.py:100-196 (``use_merged=True``): ``forward(pose (N, 69), betas)`` -> per-sample min over the mixture
    of 0.5 (x - mu)^T P (x - mu) - log(nll_weight).  ``gmm``: {'means', 'covars', 'weights'} as in gmm_08.pkl."""

    def __init__(self, gmm, device='cuda:0'):
        super().__init__()
        from .engine import gmm_constants
        self._c = gmm_constants(gmm, torch.device(device))
        self.num_gaussians = self._c['M']

    def forward(self, pose, betas=None):
        return _GMMFn.apply(pose, self)


class VPoser(nn.Module):
    """human_body_prior/models/vposer_model.py:32-106, eval mode: ``encode(pose_body (N, 63))`` -> ``Normal(mean, softplus(.))``,
    ``decode(Zin (N, 32))`` -> {'pose_body' (N, 21, 3), 'pose_body_matrot' (N, 21, 9)}.  ``state_dict``: the checkpoint's keys
    (encoder_net.{1,2,4,6,7,8.mu,8.logvar}, decoder_net.{0,3,5}); BatchNorm layers are folded into the Linear behind them."""

    def __init__(self, state_dict, device='cuda:0'):
        super().__init__()
        from .engine import fold_vposer
        self.device = torch.device(device)
        self._w = fold_vposer(state_dict, self.device)
        self._ws = torch.zeros(16 << 20, device=self.device)

    def _lin(self, rows, x, ldx, fin, w, ldw, b, fout, act):
        y = torch.empty(rows, fout, device=self.device)
        check(_lib.load().nemo_gemm_f32(0, 1, rows, fout, fin, x, ldx, dptr(w), ldw, dptr(y), fout, dptr(b), act, None, 0, 0, 1.0,
                                        0, 0, dptr(self._ws), self._ws.numel() * 4, _st()), 'nemo_gemm_f32')
        return y

    @torch.no_grad()
    def encode(self, pose_body):
        w = self._w
        x = _dev(pose_body).reshape(-1, 63)
        n = x.shape[0]
        e1 = self._lin(n, dptr(x), 63, 63, w['e2w'], 63, w['e2b'], 512, 2)
        mulv = self._lin(n, dptr(e1), 512, 512, w['emw'], 512, w['emb'], 64, 0)
        return torch.distributions.normal.Normal(mulv[:, :32], torch.nn.functional.softplus(mulv[:, 32:]))

    @torch.no_grad()
    def decode(self, Zin):
        w = self._w
        z = _dev(Zin).reshape(-1, 32)
        n = z.shape[0]
        d1 = self._lin(n, dptr(z), 32, 32, w['d0w'], 32, w['d0b'], 512, 2)
        d2 = self._lin(n, dptr(d1), 512, 512, w['d3w'], 512, w['d3b'], 512, 2)
        d3 = self._lin(n, dptr(d2), 512, 512, w['d5w'], 512, w['d5b'], 126, 0)
        R = torch.empty(n, 21, 9, device=self.device)
        aa = torch.empty(n, 63, device=self.device)
        check(_lib.load().nemo_rot6d_fwd(n, 21, dptr(d3), 126, 0, dptr(R), dptr(aa), _st()), 'nemo_rot6d_fwd')
        return {'pose_body': aa.reshape(n, 21, 3), 'pose_body_matrot': R}
