"""Loaders for the real (licensed, not redistributable) model files the reference reads from
``software/`` (hmr/hmr_config.py:70-76, nemo/neural_motion_model.py:217-238).

The benchmarks use ``synthetic.py``; these loaders exist so that a user who owns the files can run the
engine on them (file formats covered by tests/test_abi_and_host.py with synthetic content).  Every
loader fails loudly.
"""
from __future__ import annotations

import glob
import os
import pickle

import numpy as np
import torch

from .synthetic import JOINT_MAP_49, SMPL_PARENTS

# smplx/vertex_ids.py (smplx==0.1.28) 'smplh' table in VertexJointSelector order
# (nose, reye, leye, rear, lear, LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel, then the ten
# finger tips).  Model data, not arithmetic; override with NEMO_SMPL_VERTEX_IDS=path.npy if needed.
_SMPL_SELECTOR_VIDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                       2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]


def load_smpl_assets(model_dir='software/smpl', extra_path='software/spin_data/J_regressor_extra.npy'):
    cand = sorted(glob.glob(os.path.join(model_dir, '*NEUTRAL*.pkl')) + glob.glob(os.path.join(model_dir, '*.pkl')))
    if not cand:
        raise FileNotFoundError(f'no SMPL .pkl under {model_dir}; pass smpl_assets=... '
                                '(see nemo_cvpr2023_amd.synthetic.make_smpl_assets for the expected keys)')
    with open(cand[0], 'rb') as f:
        d = pickle.load(f, encoding='latin1')

    def arr(x):
        if hasattr(x, 'toarray'):
            x = x.toarray()
        if hasattr(x, 'r'):          # chumpy array
            x = x.r
        return np.asarray(x)
    nv = arr(d['v_template']).shape[0]
    vids = _SMPL_SELECTOR_VIDS
    if os.environ.get('NEMO_SMPL_VERTEX_IDS'):
        vids = np.load(os.environ['NEMO_SMPL_VERTEX_IDS']).tolist()
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    return {
        'v_template': t(arr(d['v_template'])),
        'shapedirs': t(arr(d['shapedirs'])[:, :, :10]),
        'posedirs': t(arr(d['posedirs']).reshape(nv * 3, -1).T),        # smplx: (207, 3*NV)
        'J_regressor': t(arr(d['J_regressor'])),
        'J_regressor_extra': t(np.load(extra_path)),
        'lbs_weights': t(arr(d['weights'])),
        'parents': torch.tensor(SMPL_PARENTS, dtype=torch.long),
        'extra_vids': torch.tensor(vids, dtype=torch.long),
        'joint_map': torch.tensor(JOINT_MAP_49, dtype=torch.long),
    }


def load_vposer_state(expr_dir='software/V02_05'):
    ck = sorted(glob.glob(os.path.join(expr_dir, 'snapshots', '*.ckpt')) +
                glob.glob(os.path.join(expr_dir, 'snapshots', '*.pt')))
    if not ck:
        raise FileNotFoundError(f'no VPoser snapshot under {expr_dir}/snapshots; pass vposer_state=...')
    sd = torch.load(ck[-1], map_location='cpu', weights_only=False)
    sd = sd.get('state_dict', sd)
    return {k.replace('vp_model.', ''): v for k, v in sd.items()}     # model_loader.py:64-83


def load_gmm(folder='software/spin_data', num_gaussians=8):
    path = os.path.join(folder, 'gmm_{:02d}.pkl'.format(num_gaussians))
    if not os.path.exists(path):
        raise FileNotFoundError(f'{path} not found; pass gmm={{means, covars, weights}}')
    with open(path, 'rb') as f:
        g = pickle.load(f, encoding='latin1')
    return {'means': np.asarray(g['means']), 'covars': np.asarray(g['covars']),
            'weights': np.asarray(g['weights'])}


def load_real_assets(smpl_assets=None, vposer_state=None, gmm=None):
    return (smpl_assets if smpl_assets is not None else load_smpl_assets(),
            vposer_state if vposer_state is not None else load_vposer_state(),
            gmm if gmm is not None else load_gmm())
