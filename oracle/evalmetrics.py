"""ORACLE (test infrastructure only) -- CPU restatement of the reference's evaluation numbers.

``eval_2d`` (nemo/neural_motion_model.py:522-710) and ``eval_3d`` (:1056-1282) of the reference class,
restated over ``OracleNemo`` with plain torch / numpy.  Pinned against ``tests/golden/eval_eval_v2.npz``
(written by ``tools/gen_golden.py::run_eval_case`` from the CSV files the real reference produces).
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .model import collate_gt_2d


def _labels(seqs, label_type):
    """:2908-2961 for the label types the evaluation reads ('gt', 'op', 'vibe')."""
    if label_type in ('gt', 'op'):
        return collate_gt_2d(seqs, label_type)
    gt = [np.array(seqs.sequences[v][label_type + '_joints2d']) for v in range(seqs.num_views)]
    pts = torch.tensor(np.array(gt)).float()
    d0 = pts[..., 0].max(-1)[0] - pts[..., 0].min(-1)[0]
    d1 = pts[..., 1].max(-1)[0] - pts[..., 1].min(-1)[0]
    return pts, torch.sqrt(d0 ** 2 + d1 ** 2) + 1e-4


def f_pck(pred, gt_xy, conf, gt_size):
    """:524-531: 100 * #(visible joints within 5 % of the bbox diagonal) / #(visible joints)."""
    rmse = torch.sqrt(1e-6 + ((pred - gt_xy) ** 2).sum(-1, keepdim=True))
    mask = (conf > 0.5).float()
    return 100.0 * (mask * (rmse < 0.05 * gt_size[:, None, None]).float()).sum() / mask.sum()


def frame_list(ncol, num_frames):
    """:587-588: frame_idx = round(cidx / ncol * num_frames)."""
    return [int(np.round(c / ncol * num_frames)) for c in range(ncol)]


def eval_2d(model, seqs):
    """-> dict of per-view lists: recon_error_2d-{ours,op,vibe}, pck-{ours,op,vibe}."""
    V, T = model.V, model.T
    gt_all, size_all = _labels(seqs, 'gt')
    others = {'op': _labels(seqs, 'op')[0], 'vibe': _labels(seqs, 'vibe')[0]}
    vi, fi = model.full_indices()
    with torch.no_grad():
        p = model.get_preds_batch(vi, fi)                                   # get_preds(), :568
        pts = model.learned_camera_projection(p['j'], vi).reshape(V, T, 25, 2)
    stats = {}
    frames = frame_list(T, T)
    for v in range(V):
        gt = gt_all[v, frames, :15]
        size = size_all[v, frames]
        cands = {'ours': pts[v, frames, :15]}
        cands.update({k: a[v, frames, :15, :2] for k, a in others.items()})
        for k, pred in cands.items():
            err = ops.keypoint_loss(pred, gt[..., :2], gt[..., 2:], None, 'rmse').mean()         # :636-640
            stats.setdefault('recon_error_2d-' + k, []).append(float(err))
            stats.setdefault('pck-' + k, []).append(float(f_pck(pred, gt[..., :2], gt[..., 2:], size)))
    return stats


def smpl_from_aa(model, body_aa):
    """``self.smpl(betas=None, body_pose=aa, global_orient=None, pose2rot=True)`` (:1158-1163): zero
    betas and orient, matrix-form Rodrigues of lbs.py:303-334 -> vertices, joints49[:15]."""
    n = body_aa.shape[0]
    aa = torch.cat([torch.zeros(n, 3), body_aa.reshape(n, 69).float()], 1)
    R = ops.lbs_rodrigues(aa.reshape(-1, 3)).reshape(n, 24, 3, 3)
    v, j49, _ = model.smpl.forward(torch.zeros(1, 10), R)
    return v.numpy(), j49[:, :15].numpy()


def reconstruction_error(S1, S2):
    """nemo/utils/pose_utils.py:148-160 with pa=False."""
    return float(np.sqrt(((S1 - S2) ** 2).sum(-1)).mean(-1).mean())


def dynamic_mask(j_gt, fps_mult):
    """:1081-1116: frames between the first and the last one whose fastest joint moves >= 2 m/s."""
    vel = np.sqrt(((j_gt[1:] - j_gt[:-1]) ** 2).sum(-1)) * (30 * fps_mult)
    inds = np.where(vel.max(1) >= 2)[0]
    mask = np.zeros((j_gt.shape[0],))
    mask[inds.min():inds.max()] = 1
    return mask


def eval_3d(model, seqs, dynamic_only=False):
    """-> dict of per-view lists: mpjpe-{ours,vibe}, mpvpe-{ours,vibe} (millimetres, no Procrustes)."""
    V, T = model.V, model.T
    vi, fi = model.full_indices()
    with torch.no_grad():
        poses = model.get_preds_batch(vi, fi)['poses'].reshape(V, T, 69)
    stats = {}
    for v in range(V):
        s = seqs.sequences[v]
        gt_pose = torch.stack(s['pose_3d_gt'])[:, 3:]
        frames = frame_list(T, T)
        with torch.no_grad():
            v_gt, j_gt = smpl_from_aa(model, gt_pose)
            if dynamic_only:
                mask = dynamic_mask(j_gt, seqs.framerate_multiplier[v])
                frames = [f for f in frames if mask[f] != 0]
            v_pred, j_pred = smpl_from_aa(model, poses[v])
            vibe = torch.tensor(np.array(s['pose']))[:, 3:-1]
            v_vibe, j_vibe = smpl_from_aa(model, vibe)
        for k, (vv, jj) in {'ours': (v_pred, j_pred), 'vibe': (v_vibe, j_vibe)}.items():
            stats.setdefault('mpjpe-' + k, []).append(1000 * reconstruction_error(j_gt[frames], jj[frames]))
            stats.setdefault('mpvpe-' + k, []).append(1000 * reconstruction_error(v_gt[frames], vv[frames]))
    return stats
