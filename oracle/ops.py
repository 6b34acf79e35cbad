"""ORACLE (test infrastructure only) -- operator-level CPU restatement of the NeMo hot path.

Plain, unfused PyTorch fp32 restatement of the reference's differentiable operators.
Every function cites the reference file:line it restates.  Gradients come from
``torch.autograd`` exactly as in the reference.  Parity is PINNED: each function
here is checked against golden vectors produced by importing the real reference
(``tools/gen_golden.py`` -> ``tests/golden/fn_*.npz``, ``tests/test_oracle_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product path (``nemo_cvpr2023_amd``) never does.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- rotations
def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """hmr/geometry.py:47-61.  The six numbers are read as a 3x2 matrix (interleaved
    columns); Gram-Schmidt; the result stacks b1,b2,b3 as *columns*."""
    m = x.reshape(-1, 3, 2)
    a1, a2 = m[..., 0], m[..., 1]
    b1 = F.normalize(a1, dim=1)
    proj = (b1 * a2).sum(1, keepdim=True)
    b2 = F.normalize(a2 - proj * b1, dim=1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=2)


def rotmat_to_quat(R: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """hmr/geometry.py:266-346 (== human_body_prior/tools/tgm_conversion.py:241-322).
    Four-way branch on the diagonal; quaternion is (w, x, y, z).  The reference works on
    the transpose of a 3x4 matrix; m[i][j] below is R[j][i]."""
    m = R.transpose(1, 2)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    d2 = m22 < eps
    d01 = m00 > m11
    d0n1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    t1 = 1 - m00 + m11 - m22
    t2 = 1 - m00 - m11 + m22
    t3 = 1 + m00 + m11 + m22
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], 1)
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], 1)
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], 1)
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], 1)
    c0 = (d2 & d01).to(R.dtype).unsqueeze(1)
    c1 = (d2 & ~d01).to(R.dtype).unsqueeze(1)
    c2 = (~d2 & d0n1).to(R.dtype).unsqueeze(1)
    c3 = (~d2 & ~d0n1).to(R.dtype).unsqueeze(1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    denom = torch.sqrt(t0.unsqueeze(1) * c0 + t1.unsqueeze(1) * c1 + t2.unsqueeze(1) * c2
                       + t3.unsqueeze(1) * c3)
    return 0.5 * (q / denom)


def quat_to_aa(q: torch.Tensor) -> torch.Tensor:
    """hmr/geometry.py:213-263 (== tgm_conversion.py:325-373)."""
    v = q[..., 1:]
    s2 = (v * v).sum(-1)
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    return v * k.unsqueeze(-1)


def rotmat_to_aa(R: torch.Tensor, zero_nan: bool = True) -> torch.Tensor:
    """hmr/geometry.py:181-210 (``zero_nan=True``: NaNs are overwritten with 0, :209) and
    human_body_prior/tools/rotation_tools.py:73-81 (``zero_nan=False``)."""
    aa = quat_to_aa(rotmat_to_quat(R.reshape(-1, 3, 3)))
    if zero_nan:
        aa = torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)
    return aa


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """hmr/geometry.py:25-45."""
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1).view(-1, 3, 3)


def batch_rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """hmr/geometry.py:9-23: quaternion-form Rodrigues, angle = ||theta + 1e-8||."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / angle
    half = 0.5 * angle
    return quat_to_rotmat(torch.cat([torch.cos(half), torch.sin(half) * axis], 1))


def lbs_rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """human_body_prior/body_model/lbs.py:303-334: matrix-form Rodrigues (eval path,
    ``pose2rot=True``)."""
    angle = torch.norm(theta + 1e-8, dim=1, keepdim=True)
    d = theta / angle
    z = torch.zeros_like(angle)
    K = torch.cat([z, -d[:, 2:3], d[:, 1:2], d[:, 2:3], z, -d[:, 0:1], -d[:, 1:2], d[:, 0:1], z],
                  1).view(-1, 3, 3)
    s, c = torch.sin(angle).unsqueeze(1), torch.cos(angle).unsqueeze(1)
    return torch.eye(3, dtype=theta.dtype) + s * K + (1 - c) * torch.bmm(K, K)


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """hmr/geometry.py:78-106: p' = R p + t; p'/z; K p'."""
    p = torch.einsum('bij,bkj->bki', rotation, points) + translation.unsqueeze(1)
    p = p / p[:, :, 2:3]
    f = torch.as_tensor(focal_length, dtype=p.dtype).reshape(-1, 1)
    u = f * p[:, :, 0] + camera_center[:, 0:1] * p[:, :, 2]
    v = f * p[:, :, 1] + camera_center[:, 1:2] * p[:, :, 2]
    return torch.stack([u, v], -1)


# --------------------------------------------------------------------------- SMPL
class SMPLOracle:
    """smplx.SMPL.forward (third-party, smplx==0.1.28) + hmr/smpl.py:29-43 on the
    arithmetic of human_body_prior/body_model/lbs.py:164-404."""

    def __init__(self, assets: dict):
        self.a = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in assets.items()}
        self.parents = [int(p) for p in assets['parents']]

    def shaped(self, betas):
        a = self.a
        v_shaped = a['v_template'] + torch.einsum('bl,mkl->bmk', betas, a['shapedirs'])  # lbs.py:209,299
        J = torch.einsum('bik,ji->bjk', v_shaped, a['J_regressor'])                       # lbs.py:216,274
        return v_shaped, J

    def forward(self, betas, rotmats, return_verts=True):
        """betas (1|B,10), rotmats (B,24,3,3) [global orient first] -> verts (B,V,3),
        joints49 (B,49,3), joints54."""
        a = self.a
        B = rotmats.shape[0]
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        v_shaped, J = self.shaped(betas)
        eye = torch.eye(3, dtype=rotmats.dtype)
        pose_feature = (rotmats[:, 1:] - eye).reshape(B, -1)                               # lbs.py:229
        v_posed = v_shaped + (pose_feature @ a['posedirs']).view(B, -1, 3)                 # lbs.py:232-235
        # kinematic chain, lbs.py:350-404
        rel = J.clone()
        rel[:, 1:] = J[:, 1:] - J[:, self.parents[1:]]
        chainR = [rotmats[:, 0]]
        chaint = [rel[:, 0]]
        for i in range(1, 24):
            p = self.parents[i]
            chainR.append(chainR[p] @ rotmats[:, i])
            chaint.append((chainR[p] @ rel[:, i].unsqueeze(-1)).squeeze(-1) + chaint[p])
        G_R = torch.stack(chainR, 1)            # (B,24,3,3)
        G_t = torch.stack(chaint, 1)            # (B,24,3) posed joints
        A_t = G_t - (G_R @ J.unsqueeze(-1)).squeeze(-1)   # lbs.py:399-402
        # skinning, lbs.py:238-252
        W = a['lbs_weights']
        A = torch.cat([G_R, A_t.unsqueeze(-1)], -1).reshape(B, 24, 12)
        T = (W @ A).view(B, -1, 3, 4)
        verts = (T[..., :3] @ v_posed.unsqueeze(-1)).squeeze(-1) + T[..., 3]
        joints = torch.cat([G_t, verts[:, a['extra_vids']]], 1)                            # smplx VertexJointSelector
        extra = torch.einsum('bik,ji->bjk', verts, a['J_regressor_extra'])                 # hmr/smpl.py:33-34
        j54 = torch.cat([joints, extra], 1)
        return verts, j54[:, a['joint_map']], j54                                          # hmr/smpl.py:36


# --------------------------------------------------------------------------- small modules
def monotonic_forward(shifts, scales, x):
    """monotonic_network.py:23-39.  shifts/scales (K,) or per-sample (N,K); x (N,1)."""
    sh, sc = torch.relu(shifts), torch.relu(scales)

    def npass(xx):
        return torch.sigmoid(sc * (xx - sh)).mean(-1, keepdim=True)
    y, z, o = npass(x), npass(torch.zeros_like(x)), npass(torch.ones_like(x))
    return (y - z) / (o - z + 1e-6)


RBF_KERNELS = {
    'gaussian': lambda a: torch.exp(-a.pow(2)),
    'linear': lambda a: a,
    'quadratic': lambda a: a.pow(2),
    'inverse_quadratic': lambda a: 1.0 / (1.0 + a.pow(2)),
    'multiquadric': lambda a: (1.0 + a.pow(2)).pow(0.5),
    'inverse_multiquadric': lambda a: 1.0 / (1.0 + a.pow(2)).pow(0.5),
    'spline': lambda a: a.pow(2) * torch.log(a + 1.0),
    'poisson_one': lambda a: (a - 1.0) * torch.exp(-a),
    'poisson_two': lambda a: ((a - 2.0) / 2.0) * a * torch.exp(-a),
    'matern32': lambda a: (1.0 + 3 ** 0.5 * a) * torch.exp(-3 ** 0.5 * a),
    'matern52': lambda a: (1.0 + 5 ** 0.5 * a + (5 / 3) * a.pow(2)) * torch.exp(-5 ** 0.5 * a),
}


def rbf_forward(log_sigmas, centres, x, kernel='quadratic'):
    """nemo/rbf.py:47-74: phi((x-c)^2 / exp(log_sigma)); x (N,1), centres (D,1)."""
    d = (x - centres.reshape(1, -1)).pow(2) / torch.exp(log_sigmas).unsqueeze(0)
    return RBF_KERNELS[kernel](d)


def gmof(residual, sqrt, rho=100.0):
    """nemo/utils/misc_utils.py:91-105."""
    sq = residual ** 2
    if sqrt:
        sq = torch.sqrt(sq.sum(-1)).unsqueeze(-1)
    return rho ** 2 * (sq / (sq + rho ** 2))


def keypoint_loss(pred, gt, weight, gt_size=None, loss_type='mse_robust'):
    """nemo/neural_motion_model.py:2806-2843."""
    m = (weight > 0.5).float()
    if loss_type == 'rmse':
        return m * torch.sqrt(1e-6 + ((pred - gt) ** 2).sum(-1, keepdim=True))
    if loss_type == 'rmse_resized':
        s = gt_size.unsqueeze(-1).unsqueeze(-1)
        return m * torch.sqrt(1e-6 + ((pred / s - gt / s) ** 2).sum(-1, keepdim=True))
    if loss_type == 'mse':
        return m * (pred - gt) ** 2
    if loss_type == 'rmse_robust':
        return m * gmof(pred - gt, sqrt=True)
    if loss_type == 'mse_robust':
        return m * gmof(pred - gt, sqrt=False)
    if loss_type == 'mse_robust_resized':
        s = gt_size.unsqueeze(-1).unsqueeze(-1)
        return m * gmof(pred / s * 1000 - gt / s * 1000, sqrt=False)
    raise ValueError(loss_type)


def per_view_mean_loss(loss_all, conf, view_idx):
    """nemo/neural_motion_model.py:3551-3558: for each view present, mean over ALL
    elements of (loss * conf); then mean over the views present."""
    total = 0
    views = view_idx.unique()
    for v in views:
        sel = view_idx == v
        total = total + (loss_all[sel] * conf[sel]).mean()
    return total / len(views)


# --------------------------------------------------------------------------- motion MLP
def motionnet_forward(sd: dict, prefix: str, x: torch.Tensor):
    """nemo/neural_motion_model.py:58-71,130-148 (FCNN + heads).  ``sd`` maps the
    reference's state_dict names to tensors.  Returns rot6d (N,144), trans (N,3)."""
    h = F.relu(F.linear(x, sd[prefix + 'net.net.0.weight'], sd[prefix + 'net.net.0.bias']))
    h = F.relu(F.linear(h, sd[prefix + 'net.net.2.weight'], sd[prefix + 'net.net.2.bias']))
    h = F.linear(h, sd[prefix + 'net.net.4.weight'], sd[prefix + 'net.net.4.bias'])
    z = F.relu(h)
    rot6d = F.linear(z, sd[prefix + 'rot_out.weight'], sd[prefix + 'rot_out.bias'])
    trans = F.linear(z, sd[prefix + 'linear_out.weight'], sd[prefix + 'linear_out.bias'])
    return rot6d, trans


# --------------------------------------------------------------------------- VPoser
class VPoserOracle:
    """human_body_prior/models/vposer_model.py:59-106 in eval mode (BatchNorm uses running
    statistics, Dropout is the identity)."""

    def __init__(self, sd: dict):
        self.sd = {k: v.clone().float() for k, v in sd.items() if v.dtype.is_floating_point}
        self.latentD = self.sd['encoder_net.8.mu.weight'].shape[0]

    def _bn(self, x, p):
        sd = self.sd
        return (x - sd[p + '.running_mean']) / torch.sqrt(sd[p + '.running_var'] + 1e-5) \
            * sd[p + '.weight'] + sd[p + '.bias']

    def _lin(self, x, p):
        return F.linear(x, self.sd[p + '.weight'], self.sd[p + '.bias'])

    def encode(self, pose_body):
        h = self._bn(pose_body.reshape(pose_body.shape[0], -1), 'encoder_net.1')
        h = F.leaky_relu(self._lin(h, 'encoder_net.2'))
        h = self._bn(h, 'encoder_net.4')
        h = self._lin(self._lin(h, 'encoder_net.6'), 'encoder_net.7')
        return self._lin(h, 'encoder_net.8.mu'), F.softplus(self._lin(h, 'encoder_net.8.logvar'))

    def decode(self, z):
        h = F.leaky_relu(self._lin(z, 'decoder_net.0'))
        h = F.leaky_relu(self._lin(h, 'decoder_net.3'))
        R = rot6d_to_rotmat(self._lin(h, 'decoder_net.5'))   # ContinousRotReprDecoder :32-45
        aa = rotmat_to_aa(R, zero_nan=False)                  # matrot2aa, rotation_tools.py:73-81
        return aa.view(z.shape[0], -1, 3), R.reshape(z.shape[0], -1, 9)


def kl_to_std_normal(mean, scale):
    """torch.distributions.kl.kl_divergence(Normal(mean, scale), Normal(0,1)) summed over the
    latent and averaged over the batch (nemo/neural_motion_model.py:2795-2802)."""
    var_ratio = scale.pow(2)
    t1 = mean.pow(2)
    return (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(1).mean()


# --------------------------------------------------------------------------- GMM prior
class GMMPriorOracle:
    """hmr/smplify/prior.py:100-196 (MaxMixturePrior, merged likelihood)."""

    def __init__(self, gmm: dict):
        means = gmm['means'].astype(np.float32)
        covs = gmm['covars'].astype(np.float32)
        precisions = np.stack([np.linalg.inv(c) for c in covs]).astype(np.float32)     # :142-143
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in gmm['covars']])            # :149-150
        const = (2 * np.pi) ** (69 / 2.0)
        nll_weights = np.asarray(gmm['weights'] / (const * (sqrdets / sqrdets.min())))    # :153-154
        self.means = torch.tensor(means)
        self.precisions = torch.tensor(precisions)
        self.nll_weights = torch.tensor(nll_weights, dtype=torch.float32).unsqueeze(0)

    def __call__(self, pose):
        d = pose.unsqueeze(1) - self.means                                                 # :182
        pd = torch.einsum('mij,bmj->bmi', self.precisions, d)                              # :184-185
        ll = 0.5 * (pd * d).sum(-1) - torch.log(self.nll_weights)                          # :186-189
        return ll.min(dim=1)[0]                                                            # :195


def joints3d_smooth_loss(joints):
    """humor/humor/fitting/fitting_loss.py:366-370 (`FittingLoss.joints3d_smooth_loss`): 0.5 * sum over consecutive frames of the
    squared joint displacement; joints (B, T, J, 3).  Pinned by tests/golden/fn_joints3d_smooth_loss.npz (the reference's own method,
    tools/gen_golden.py::run_smooth_case)."""
    return 0.5 * ((joints[:, 1:] - joints[:, :-1]) ** 2).sum()
