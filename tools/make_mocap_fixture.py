#!/usr/bin/env python3
"""Writes the tiny NeMo-MoCap-shaped dataset under tests/golden/mocap_fixture/ that pins the data layer
(nemo_cvpr2023_amd/multi_view_sequence.py vs the reference's MultiViewSequence, run on the same files by
tools/gen_golden.py --only loader).  Two "videos" of 9 and 7 frames with different image sizes, two VIBE
tracks each (partial frame coverage; the right one is NOT always the first key), one frame without an
OpenPose detection.  Everything is synthetic and seeded."""
import json
import os
import os.path as osp

import joblib
import numpy as np
import torch
from PIL import Image

ROOT = osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), 'tests', 'golden', 'mocap_fixture')
VIDEOS = [('baseball_pitch.0.mp4', 9, (12, 16)), ('baseball_pitch.1.mp4', 7, (10, 20))]   # name, frames, (H, W)


def main():
    rng = np.random.default_rng(20231)
    exp = osp.join(ROOT, 'exps')
    os.makedirs(osp.join(ROOT, 'data', 'mocap'), exist_ok=True)
    torch.save((torch.zeros(9), 1000.0), osp.join(ROOT, 'data', 'opt_cam_IMG_6289.pt'))
    for vi, (name, n, (h, w)) in enumerate(VIDEOS):
        img_dir, stem = osp.join(exp, name), name[:-4]
        for d in (img_dir, img_dir + '_openpose', img_dir + '_gt_new', osp.join(exp, stem)):
            os.makedirs(d, exist_ok=True)
        centre = np.array([300.0 + 200 * vi, 250.0])
        for t in range(n):
            Image.fromarray(np.full((h, w, 3), 10 * t, dtype=np.uint8)).save(osp.join(img_dir, f'{t + 1:06d}.png'))
            gt = (centre + 40 * rng.standard_normal((1, 17, 2)) + 5 * t).astype(np.float64)
            joblib.dump(gt, osp.join(img_dir + '_gt_new', f'{t + 1:06d}_keypoints.pkl'))
            people = []
            if not (vi == 1 and t == 3):                       # one frame without a detection
                kp = np.concatenate([gt[0, :15] + rng.standard_normal((15, 2)), 600 * rng.uniform(size=(10, 2))])
                kp = np.concatenate([kp, rng.uniform(0.1, 1.0, (25, 1))], 1)
                people.append({'pose_keypoints_2d': [round(float(x), 4) for x in kp.reshape(-1)]})
            with open(osp.join(img_dir + '_openpose', f'{t + 1:06d}_keypoints.json'), 'w') as f:
                json.dump({'version': 1.3, 'people': people}, f)
        tracks = {}
        for pid, (off, ids) in enumerate([((400.0, -300.0), np.arange(0, n)),            # far away, full coverage
                                          ((6.0, -4.0), np.arange(1, n - 1))], start=1):  # on the subject, partial
            m = len(ids)
            tracks[pid] = {
                'pose': (0.3 * rng.standard_normal((m, 72))).astype(np.float32),
                'betas': (0.1 * rng.standard_normal((m, 10))).astype(np.float32),
                'frame_ids': ids,
                'orig_cam': rng.uniform(0.5, 1.0, (m, 4)).astype(np.float32),
                'verts': (0.2 * rng.standard_normal((m, 20, 3))).astype(np.float32),
                'joints2d_img_coord': (centre + np.array(off) + 30 * rng.standard_normal((m, 49, 2))).astype(np.float32),
                'joints3d': None,
            }
        if vi == 1:                                            # the right track is not always key 2
            tracks = {1: tracks[2], 2: tracks[1]}
        joblib.dump(tracks, osp.join(exp, stem, 'vibe_output.pkl'))
        joblib.dump({'fullpose': (0.3 * rng.standard_normal((n, 156))).astype(np.float32),
                     'trans': rng.standard_normal((n, 3)).astype(np.float32)},
                    osp.join(ROOT, 'data', 'mocap', stem + '.pkl'))
    with open(osp.join(ROOT, 'cfg.json'), 'w') as f:
        json.dump({'exp_dir': 'exps', 'videos': {'names': [v[0] for v in VIDEOS]}}, f)
    print('wrote', ROOT, sum(len(fs) for _, _, fs in os.walk(ROOT)), 'files,',
          sum(osp.getsize(osp.join(d, f)) for d, _, fs in os.walk(ROOT) for f in fs) // 1024, 'KB')


if __name__ == '__main__':
    main()
