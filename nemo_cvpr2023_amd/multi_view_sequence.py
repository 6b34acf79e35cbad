"""Array-backed data layer: what ``nemo/multi_view_sequence.py:250-483``
(``MultiViewSequence``) hands to the model, without images, OpenCV, HMR or the renderer.

``load_nemo_mocap(cfg, start_phase, num_frames)`` reads the same on-disk layout the reference reads

    <exp_dir>/<name>/NNNNNN.png                      frame count (and image size of frame 1)
    <exp_dir>/<name>_openpose/NNNNNN_keypoints.json  OpenPose BODY_25, ``people[0].pose_keypoints_2d``
    <exp_dir>/<name>_gt_new/NNNNNN_keypoints.pkl     2-D ground truth, joblib pickle, array (1, >=15, 2)
    <exp_dir>/<name minus '.mp4'>/vibe_output.pkl    VIBE tracks {person_id: {'pose','frame_ids','joints2d_img_coord',...}}
    <mocap_root>/<name minus '.mp4'>.pkl             3-D ground truth {'fullpose' (F, >=66), 'trans' (F, 3)}

and produces the duck type the model needs: ``num_views, num_frames, IMG_D0, IMG_D1,
framerate_multiplier, sequences[v][key] = list over frames`` with keys ``pose_2d_op (25,3)``,
``pose_2d_gt (25,3)``, ``pose (73,)``, ``vibe_joints2d``, ``vibe_mask``, ``pose_3d_gt (72,)``,
``trans_3d_gt (3,)``.  Frame resampling ``tidx = floor(phase * n_seq_frames)`` (:411-414), number of
frames ``min(num_frames, shortest_video - 1)`` (:268-270), VIBE person selection = the track whose
mean of the first 15 joints is closest to the ground-truth 2-D centre (:60-89), tracks scattered to
full length with zeros and the valid flag appended as ONES for every frame (:30-57, :329-334 -- sic).
"""
import json
import os
import os.path as osp

import numpy as np


class ArrayMultiViewSequence:
    """Container with the attribute surface of the reference class; fill ``sequences`` with per-view dicts
    of per-frame lists (or use ``from_arrays``)."""

    def __init__(self, num_frames, img_d0, img_d1):
        self.num_frames = int(num_frames)
        self.IMG_D0, self.IMG_D1 = int(img_d0), int(img_d1)
        self.sequences = []
        self.framerate_multiplier = []

    @property
    def num_views(self):
        return len(self.sequences)

    @classmethod
    def from_arrays(cls, pose_2d_op, pose, img_d0, img_d1, pose_2d_gt=None, **extra):
        """pose_2d_op (V,T,25,3), pose (V,T,73); optional pose_2d_gt and any further (V,T,...) arrays."""
        pose_2d_op = np.asarray(pose_2d_op)
        V, T = pose_2d_op.shape[:2]
        out = cls(T, img_d0, img_d1)
        fields = dict(pose_2d_op=pose_2d_op, pose=np.asarray(pose), **extra)
        if pose_2d_gt is not None:
            fields['pose_2d_gt'] = np.asarray(pose_2d_gt)
        for v in range(V):
            out.sequences.append({k: [a[v][t] for t in range(T)] for k, a in fields.items()})
            out.framerate_multiplier.append(1.0)
        return out

    def get_image(self, v, t):
        raise NotImplementedError('the array-backed data layer carries no images (rendering is out of scope)')


def _scatter_track(person, n_frames):
    """:30-49: every per-frame array of a VIBE track scattered to the full clip length (zeros elsewhere)
    plus ``mask`` = 1 on the frames the track covers."""
    ids = np.asarray(person['frame_ids'])
    out = {}
    for k, v in person.items():
        if k in ('betas', 'frame_ids'):
            out[k] = v
            continue
        if v is None:
            continue
        full = np.zeros([n_frames] + list(v.shape[1:]), dtype=np.float32)
        full[ids] = v
        out[k] = full
    mask = np.zeros(n_frames, dtype=np.float32)
    mask[ids] = 1
    out['mask'] = mask
    return out


def _select_person(tracks, all_gt_2d):
    """:60-89: masked mean distance between the track's joint centre and the ground-truth centre."""
    best, best_d = None, np.inf
    gt_c = all_gt_2d.mean(1)
    for key, p in tracks.items():
        j2d = p['joints2d_img_coord'] if 'joints2d_img_coord' in p else p['smpl_joints2d']
        c = j2d[:, :15].mean(1)
        d = (np.sqrt(((c - gt_c) ** 2).sum(-1)) * p['mask']).sum() / p['mask'].sum()
        if best is None or d < best_d:
            best, best_d = key, d
    return tracks[best]


def _image_size(img_dir):
    from PIL import Image
    first = sorted(f for f in os.listdir(img_dir) if f.endswith('.png'))[0]
    with Image.open(osp.join(img_dir, first)) as im:
        w, h = im.size
    return h, w


def load_nemo_mocap(cfg, start_phase=0, num_frames=1000000, mocap_root='data/mocap', img_size=None):
    """cfg = the parsed ``nemo/config/*.yml`` ({'exp_dir': ..., 'videos': {'names': [...]}}).
    img_size=(H, W) skips reading the first frame of every video for its size."""
    import joblib
    import torch
    if start_phase != 0:
        raise ValueError('start_phase must be 0 (multi_view_sequence.py:298)')
    names = cfg['videos']['names']
    exp = cfg['exp_dir']
    shortest = min(len(os.listdir(osp.join(exp, n))) for n in names)                 # :262-266 (all entries)
    T = int(min(num_frames, shortest - np.round(shortest * start_phase) - 1))        # :268-270
    sizes = []
    seqs = ArrayMultiViewSequence(T, 0, 0)
    for name in names:
        img_dir = osp.join(exp, name)
        n_seq = len([f for f in os.listdir(img_dir) if f.endswith('.png')])
        seqs.framerate_multiplier.append(n_seq / T)
        sizes.append(tuple(img_size) if img_size is not None else _image_size(img_dir))
        gt_dir, op_dir, stem = img_dir + '_gt_new', img_dir + '_openpose', name[:-4]

        def gt2d(tidx):
            return np.asarray(joblib.load(osp.join(gt_dir, f'{tidx + 1:06d}_keypoints.pkl')))[0, :15]

        all_gt = np.array([gt2d(t) for t in range(n_seq)])
        tracks = joblib.load(osp.join(exp, stem, 'vibe_output.pkl'))
        tracks = {k: _scatter_track(p, n_seq) for k, p in tracks.items()}
        person = _select_person(tracks, all_gt)
        pose = person.get('pose')
        if pose is None:
            pose73 = np.zeros((n_seq, 73))
        else:
            pose73 = np.concatenate([pose, np.ones((n_seq, 1))], 1)                  # :333-334
        gt3d = joblib.load(osp.join(mocap_root, stem + '.pkl'))
        p3 = torch.tensor(np.asarray(gt3d['fullpose'])[:, :66]).float()
        p3 = torch.cat([p3, torch.zeros(p3.shape[0], 6)], 1)                         # :339-343
        t3 = torch.tensor(np.asarray(gt3d['trans'])).float()
        cur = {k: [] for k in ('pose_2d_op', 'pose_2d_gt', 'pose', 'vibe_mask', 'vibe_joints2d',
                               'pose_3d_gt', 'trans_3d_gt')}
        for f in range(T):
            phase = start_phase + (1 - start_phase) * float(f / T)
            tidx = int(np.floor(phase * n_seq))                                      # :411-414
            with open(osp.join(op_dir, f'{tidx + 1:06d}_keypoints.json')) as fh:
                people = json.load(fh)['people']
            if len(people) == 1:
                op = np.array(people[0]['pose_keypoints_2d']).reshape(25, 3)
            elif len(people) == 0:
                op = np.zeros((25, 3))                                               # :422-423
            else:
                raise ValueError(f'{name} frame {tidx + 1}: {len(people)} OpenPose detections (:424-425)')
            cur['pose_2d_op'].append(op)
            g = np.vstack([np.hstack([gt2d(tidx), np.ones((15, 1))]), np.zeros((10, 3))])   # :432-434
            cur['pose_2d_gt'].append(g)
            cur['pose'].append(pose73[tidx])
            cur['vibe_mask'].append(person['mask'][tidx])
            cur['vibe_joints2d'].append(person['joints2d_img_coord'][tidx])
            cur['pose_3d_gt'].append(p3[tidx])
            cur['trans_3d_gt'].append(t3[tidx])
        seqs.sequences.append(cur)
    seqs.IMG_D0 = max(s[0] for s in sizes)                                           # :476-481
    seqs.IMG_D1 = max(s[1] for s in sizes)
    return seqs
