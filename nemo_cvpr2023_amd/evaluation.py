"""Evaluation numbers of a fit on the device: ``eval_2d`` (PCK@0.05 bbox diagonal and
2-D reconstruction RMSE over the first 15 OpenPose joints, nemo/neural_motion_model.py:522-710) and
``eval_3d`` (MPJPE / MPVPE in millimetres without Procrustes alignment, optionally restricted to the
"dynamic range" of the ground-truth motion, :1056-1282), for the fitted model and for the OpenPose / VIBE
inputs.  As shipped, the reference's ``eval_3d`` also indexes 'vs_pose' / 'pare_pose' / 'glamr_pose',
which its own loader no longer fills (it raises on the public data); only the columns that exist are
produced here.  CSV layout = ``pandas.DataFrame(stats).to_csv`` (index column first).

SMPL evaluation (``self.smpl(betas=None, body_pose=aa, global_orient=None, pose2rot=True)``, :1158-1163)
runs through the same HIP entry points as the fit: nemo_rodrigues_fwd (matrix form of lbs.py:303-334),
nemo_fk_fwd, the blend-shape GEMM and nemo_skin_vertices; the metric reductions are a handful of torch
ops on device tensors.
"""
import csv
import os

import numpy as np
import torch

from ._lib import check, dptr
from .neural_motion_model import collate_gt_2d


def _stream():
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def frame_list(ncol, num_frames):
    """:587-588: frame_idx = round(cidx / ncol * num_frames)."""
    return [int(np.round(c / ncol * num_frames)) for c in range(ncol)]


def _views(model, num_views, view_idxs):
    if view_idxs:
        return list(view_idxs)
    n = model.num_views if num_views < 0 else min(model.num_views, num_views)
    return list(range(n))


def _write_csv(path, stats):
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    keys = list(stats)
    with open(path, 'w', newline='') as f:
        wr = csv.writer(f)
        wr.writerow([''] + keys)
        for i in range(len(stats[keys[0]]) if keys else 0):
            wr.writerow([i] + [stats[k][i] for k in keys])


def f_pck(pred, gt_xy, conf, gt_size):
    """:524-531."""
    rmse = torch.sqrt(1e-6 + ((pred - gt_xy) ** 2).sum(-1, keepdim=True))
    mask = (conf > 0.5).float()
    return 100.0 * (mask * (rmse < 0.05 * gt_size[:, None, None]).float()).sum() / mask.sum()


@torch.no_grad()
def eval_2d(model, out_dir=None, num_frames=-1, num_views=-1, view_idxs=()):
    seqs, dev = model.multi_view_seqs, model.device
    T = model.num_frames
    ncol = T if num_frames < 0 else min(T, num_frames)
    gt_all, size_all = (t.to(dev) for t in collate_gt_2d(seqs, 'gt'))
    others = {}
    for k in ('op', 'vibe'):
        try:
            others[k] = collate_gt_2d(seqs, k)[0].to(dev)
        except KeyError:
            pass
    p = model.get_preds()                                                   # :568
    V = model.num_views
    vi = p['view_idx'].reshape(-1)
    pts = model.learned_camera_projection(p['j'].reshape(-1, 25, 3), vi).reshape(V, T, 25, 2)
    stats = {}
    frames = torch.tensor(frame_list(ncol, T), device=dev)
    for v in _views(model, num_views, view_idxs):
        gt, size = gt_all[v, frames, :15], size_all[v, frames]
        cands = {'ours': pts[v, frames, :15]}
        cands.update({k: a[v, frames, :15, :2] for k, a in others.items()})
        for k, pred in cands.items():
            err = model.keypoint_loss(pred, gt[..., :2], gt[..., 2:], loss_type='rmse').mean()
            stats.setdefault('recon_error_2d-' + k, []).append(float(err))
            stats.setdefault('pck-' + k, []).append(float(f_pck(pred, gt[..., :2], gt[..., 2:], size)))
    if out_dir:
        _write_csv(os.path.join(out_dir, 'eval_2d.csv'), stats)
    return stats


@torch.no_grad()
def smpl_from_aa(model, body_aa, chunk=2048):
    """(n, 69) axis-angle body poses -> vertices (n, NV, 3), joints (n, 15, 3) = first 15 of the 49-joint map,
    zero global orientation, zero betas, no translation."""
    e = model.engine
    ctx, L, dev = e.ctx, e.lib, model.device
    n = body_aa.shape[0]
    zero_betas = not bool(torch.any(e.betas != 0))
    if not zero_betas:                                      # eval always uses betas = None (zeros)
        ctx.set_betas(np.zeros(10, dtype=np.float32))
        ctx._betas_version = None                           # (restored below, before returning)
    theta = torch.zeros(n, 72, device=dev)
    theta[:, 3:] = body_aa.to(dev, torch.float32).reshape(n, 69)
    R = torch.empty(n, 24, 9, device=dev)
    check(L.nemo_rodrigues_fwd(n * 24, dptr(theta), 1, dptr(R), _stream()), 'nemo_rodrigues_fwd')
    A, Jp, PF = (torch.empty(n, *s, device=dev) for s in ((24, 12), (24, 3), (208,)))
    check(L.nemo_fk_fwd(ctx.handle, n, dptr(R), dptr(A), dptr(Jp), dptr(PF), 208, _stream()), 'nemo_fk_fwd')
    NV3, ldP = 3 * e.NV, ctx.ldP
    verts = torch.empty(n, e.NV, 3, device=dev)
    VP = torch.empty(min(n, chunk), ldP, device=dev)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        e.gemm(0, 0, m, NV3, 207, PF.data_ptr() + 4 * c0 * 208, 208, ctx.posedirs, ldP, dptr(VP), ldP,
               bias=ctx.v_shaped)
        check(L.nemo_skin_vertices(ctx.handle, m, dptr(VP), ldP, A.data_ptr() + 4 * c0 * 288, None, 3,
                                   verts.data_ptr() + 4 * c0 * NV3, _stream()), 'nemo_skin_vertices')
    vids = [int(x) for x in e._assets['extra_vids']]
    cols = []
    for idx in e._jm[:15]:                                   # 49-joint map -> SMPL-54 index
        if idx < 24:
            cols.append(Jp[:, idx])
        elif idx < 45:
            cols.append(verts[:, vids[idx - 24]])
        else:
            raise NotImplementedError('regressor joints are not among the first 15 of the map')
    joints = torch.stack(cols, 1)
    if not zero_betas:
        e.sync_betas()            # the learned betas go back into the context (captured step graphs read it)
    return verts, joints


def _recon_error(a, b):
    """nemo/utils/pose_utils.py:148-160 with pa=False."""
    return float(torch.sqrt(((a - b) ** 2).sum(-1)).mean(-1).mean())


def dynamic_mask(j_gt, fps_mult):
    """:1081-1116: frames between the first and the last one whose fastest joint moves >= 2 m/s."""
    vel = torch.sqrt(((j_gt[1:] - j_gt[:-1]) ** 2).sum(-1)) * (30 * fps_mult)
    inds = torch.nonzero(vel.max(1)[0] >= 2).reshape(-1)
    mask = torch.zeros(j_gt.shape[0], dtype=torch.bool, device=j_gt.device)
    if inds.numel():
        mask[int(inds.min()):int(inds.max())] = True
    return mask


@torch.no_grad()
def eval_3d(model, out_dir=None, num_frames=-1, num_views=-1, view_idxs=(), dynamic_only=False):
    seqs, dev = model.multi_view_seqs, model.device
    T = model.num_frames
    ncol = T if num_frames < 0 else min(T, num_frames)
    poses = model.get_preds_batch(*model.full_indices(), with_vertices=False)['poses'].reshape(
        model.num_views, T, 69)
    stats = {}
    for v in _views(model, num_views, view_idxs):
        s = seqs.sequences[v]
        frames = torch.tensor(frame_list(ncol, T), device=dev)
        v_gt, j_gt = smpl_from_aa(model, torch.stack(list(s['pose_3d_gt']))[:, 3:])
        if dynamic_only:
            mask = dynamic_mask(j_gt, seqs.framerate_multiplier[v])
            frames = frames[mask[frames]]
        cands = {'ours': smpl_from_aa(model, poses[v]),
                 'vibe': smpl_from_aa(model, torch.tensor(np.array(s['pose']))[:, 3:-1])}
        for k, (vv, jj) in cands.items():
            stats.setdefault('mpjpe-' + k, []).append(1000 * _recon_error(j_gt[frames], jj[frames]))
            stats.setdefault('mpvpe-' + k, []).append(1000 * _recon_error(v_gt[frames], vv[frames]))
    if out_dir:
        _write_csv(os.path.join(out_dir, 'eval_3d_dynamic.csv' if dynamic_only else 'eval_3d.csv'), stats)
    return stats


def evaluate_all(model, out_dir=None):
    """scripts/learned_multi_view_recon_nn.py:333-335.  Sequences without 3-D ground truth skip eval_3d."""
    res = {'eval_2d': None, 'eval_3d': None, 'eval_3d_dynamic': None}
    s0 = model.multi_view_seqs.sequences[0]
    if 'pose_2d_gt' in s0:
        res['eval_2d'] = eval_2d(model, out_dir)
    if 'pose_3d_gt' in s0:
        res['eval_3d'] = eval_3d(model, out_dir)
        res['eval_3d_dynamic'] = eval_3d(model, out_dir, dynamic_only=True)
    return res
