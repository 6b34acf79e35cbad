"""Seeded synthetic stand-ins for the licensed assets and the dataset.

The real SMPL / VPoser / GMM / J_regressor_extra files are not redistributable
(reference loads them from ``software/`` -- hmr/hmr_config.py:70-76,
nemo/neural_motion_model.py:217-238), so every test and benchmark in this
repository runs on SMPL-*shaped* random assets produced here.  The same
generator feeds (a) the golden-vector script that drives the real reference,
(b) the oracle and (c) the HIP product path, so all three see identical
numbers.  Shapes and distributions follow SURVEY.md section 8(d).

All draws come from explicit CPU ``torch.Generator`` objects, so the values are
identical on the build container and on the GPU box.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

# SMPL kinematic tree (24 joints); parents[i] < i for i >= 1.
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16,
                17, 18, 19, 20, 21]

# 49-entry joint map = [JOINT_MAP[n] for n in JOINT_NAMES] of
# hmr/hmr_constants.py:37-150 (model data, verified against the reference by
# tools/gen_golden.py).  Indices refer to the 54-joint superset
# [24 FK joints | 21 selector vertices | 9 J_regressor_extra rows].
JOINT_MAP_49 = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26,
                27, 28, 29, 30, 31, 32, 33, 34,
                8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51,
                52, 53, 24, 26, 25, 28, 27]

NUM_SELECTOR_VERTS = 21       # smplx VertexJointSelector for SMPL (joints 24..44)
NUM_EXTRA_ROWS = 9            # hmr/smpl.py:22-26 (J_regressor_extra)
FOCAL_LENGTH = 5000.0         # hmr/hmr_constants.py:1


def _row_normalised(gen, rows, cols, power):
    w = torch.rand(rows, cols, generator=gen, dtype=torch.float64) ** power
    return (w / w.sum(1, keepdim=True)).float()


def _locality_assets(nv: int, g, skin_nnz: int) -> dict:
    """A body model with the published SMPL model's SPATIAL structure (the model file itself is not redistributable,
    hmr/hmr_config.py:70-76): a skeleton of 24 rest joints laid out along the kinematic tree; vertices stored body part by
    body part (consecutive indices share their dominant joint, as SMPL's do within a part), each around its part's bone;
    1 - 2 dominant skinning weights -- the part's joint and its parent or a child, a rest among their neighbours in the tree --;
    J_regressor rows that are sparse and local (a joint is regressed from ~1 % of the vertices, those of its own and its
    parent's part); pose blend shapes that are large for the joints next to the vertex and 50 x smaller elsewhere.
    Everything the kernels' access patterns depend on (which joints a 16-vertex tile touches, how many regressor entries a
    joint has) then looks like the real model's instead of a random permutation's."""
    par = SMPL_PARENTS
    # rest skeleton: every joint 8 - 25 cm from its parent, in a direction that depends on the limb
    J = torch.zeros(24, 3)
    for j in range(1, 24):
        d = torch.randn(3, generator=g)
        J[j] = J[par[j]] + d / d.norm() * (0.08 + 0.17 * float(torch.rand((), generator=g)))
    # vertices per part ~ bone length (at least 64), parts in joint order -> contiguous index ranges
    w_part = torch.tensor([0.12] + [float((J[j] - J[par[j]]).norm()) for j in range(1, 24)])
    cnt = torch.clamp((w_part / w_part.sum() * nv).long(), min=min(64, nv // 24))
    cnt[0] += nv - int(cnt.sum())
    if int(cnt[0]) < 1:
        cnt = torch.full((24,), nv // 24, dtype=torch.long)
        cnt[0] += nv - int(cnt.sum())
    part = torch.repeat_interleave(torch.arange(24), cnt)
    children = [[c for c in range(24) if par[c] == j] for j in range(24)]
    vt = torch.zeros(nv, 3)
    W = torch.zeros(nv, 24, dtype=torch.float64)
    t_along = torch.rand(nv, generator=g)                       # position along the part's bone (towards the parent)
    off = 0.04 * torch.randn(nv, 3, generator=g)
    r1, r2 = torch.rand(nv, generator=g), torch.rand(nv, generator=g)
    for v in range(nv):
        j = int(part[v])
        p = par[j] if j > 0 else 0
        vt[v] = J[j] + float(t_along[v]) * 0.6 * (J[p] - J[j]) + off[v]
        # second joint: the parent for vertices on the parent's side of the bone, else a child (if any)
        second = p if (float(t_along[v]) > 0.5 or not children[j]) else children[j][int(r1[v] * len(children[j])) % len(children[j])]
        main = 0.55 + 0.45 * abs(2.0 * float(t_along[v]) - 1.0) ** 0.5          # 1 - 2 dominant weights
        W[v, j] += main
        if second != j:
            W[v, second] += (1.0 - main) * 0.85
        rest = 1.0 - float(W[v].sum())
        if skin_nnz >= 3 and rest > 0:
            nb = [x for x in ([par[p]] if p > 0 else []) + children[second if second != j else j] if x not in (j, second) and x >= 0]
            nb = nb[:max(skin_nnz - 2, 0)] if skin_nnz < 24 else nb
            if nb:
                for x in nb:
                    W[v, x] += rest / len(nb)
        W[v] /= W[v].sum()
    if skin_nnz < 24:
        keep = torch.zeros_like(W).scatter_(1, W.topk(int(skin_nnz), dim=1).indices, 1.0)
        W = W * keep
        W = W / W.sum(1, keepdim=True)
    # J_regressor: joint j from ~nv / 100 vertices of its own and its parent's part (positive, rows sum to one)
    starts = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.long), cnt]), 0)
    Jr = torch.zeros(24, nv, dtype=torch.float64)
    k_reg = max(8, nv // 100)
    for j in range(24):
        cand = torch.cat([torch.arange(int(starts[q]), int(starts[q + 1])) for q in {j, par[j] if j > 0 else 0}])
        pick = cand[torch.randperm(cand.numel(), generator=g)[:min(k_reg, cand.numel())]]
        wts = torch.rand(pick.numel(), generator=g, dtype=torch.float64) ** 2 + 1e-3
        Jr[j, pick] = wts / wts.sum()
    Jx = torch.zeros(NUM_EXTRA_ROWS, nv, dtype=torch.float64)
    for q in range(NUM_EXTRA_ROWS):
        j = int(torch.randint(0, 24, (1,), generator=g))
        cand = torch.arange(int(starts[j]), int(starts[j + 1]))
        pick = cand[torch.randperm(cand.numel(), generator=g)[:min(max(4, k_reg // 4), cand.numel())]]
        wts = torch.rand(pick.numel(), generator=g, dtype=torch.float64) + 1e-3
        Jx[q, pick] = wts / wts.sum()
    # pose blend shapes: feature k belongs to joint 1 + k // 9; large where that joint is the vertex's part, its parent or a child
    P = 1e-3 * torch.randn(207, nv, 3, generator=g)
    near = torch.zeros(24, 24, dtype=torch.bool)
    for j in range(24):
        near[j, j] = True
        if j > 0:
            near[j, par[j]] = near[par[j], j] = True
    scale = torch.where(near[1 + torch.arange(207) // 9][:, part], torch.tensor(1.0), torch.tensor(0.02))     # (207, nv)
    P = (P * scale.unsqueeze(-1)).reshape(207, nv * 3)
    return dict(v_template=vt, lbs_weights=W.float(), J_regressor=Jr.float(), J_regressor_extra=Jx.float(), posedirs=P,
                shapedirs=0.01 * torch.randn(nv, 3, 10, generator=g))


def make_smpl_assets(num_verts: int = 6890, seed: int = 1, skin_nnz: int = 24, locality: bool = False) -> dict:
    """SMPL-shaped random model.  Keys mirror the buffers smplx.SMPL registers.

    ``locality`` (round 6): the spatially structured model of ``_locality_assets`` -- vertices ordered by body part, 1 - 2 dominant
    skinning weights, sparse local joint regressors -- instead of the default's random permutation (every golden fixture and
    profile of rounds 1 - 5 is on the default; bench.py's `locality_body_model` leg and its counters are on this one).

    ``skin_nnz``: non-zero skinning weights per vertex.  24 (every joint; what the committed golden fixtures were
    generated with) or fewer: the published SMPL model file has at most FOUR non-zero ``weights`` per vertex (Loper et
    al. 2015, sec. 3: the blend weights are kept sparse for compatibility with rendering engines), which is what
    ``skin_nnz=4`` reproduces -- each vertex keeps its ``skin_nnz`` largest random weights, renormalised."""
    g = torch.Generator().manual_seed(seed)
    nv = num_verts
    if locality:
        a = _locality_assets(nv, g, skin_nnz)
        a['parents'] = torch.tensor(SMPL_PARENTS, dtype=torch.long)
        # the selector vertices of smplx (nose, eyes, ears, toes, heels, finger tips): extremities -> vertices of leaf parts
        a['extra_vids'] = torch.randint(0, nv, (NUM_SELECTOR_VERTS,), generator=g)
        a['joint_map'] = torch.tensor(JOINT_MAP_49, dtype=torch.long)
        return a
    a = {}
    a['v_template'] = 0.3 * torch.randn(nv, 3, generator=g)
    a['shapedirs'] = 0.01 * torch.randn(nv, 3, 10, generator=g)
    a['posedirs'] = 1e-3 * torch.randn(207, nv * 3, generator=g)
    a['J_regressor'] = _row_normalised(g, 24, nv, 8)
    a['J_regressor_extra'] = _row_normalised(g, NUM_EXTRA_ROWS, nv, 8)
    a['lbs_weights'] = _row_normalised(g, nv, 24, 6)
    if skin_nnz < 24:
        w = a['lbs_weights'].double()
        keep = torch.zeros_like(w).scatter_(1, w.topk(int(skin_nnz), dim=1).indices, 1.0)
        w = w * keep
        a['lbs_weights'] = (w / w.sum(1, keepdim=True)).float()
    a['parents'] = torch.tensor(SMPL_PARENTS, dtype=torch.long)
    a['extra_vids'] = torch.randint(0, nv, (NUM_SELECTOR_VERTS,), generator=g)
    a['joint_map'] = torch.tensor(JOINT_MAP_49, dtype=torch.long)
    return a


def make_vposer_state(seed: int = 3, num_neurons: int = 512, latent: int = 32) -> dict:
    """Random VPoser-v2 weights with the checkpoint key names of
    human_body_prior/models/vposer_model.py:69-88 (encoder_net.{1,2,4,6,7,8},
    decoder_net.{0,3,5}).  BatchNorm running statistics are non-trivial so the
    eval-mode affine is exercised."""
    g = torch.Generator().manual_seed(seed)

    def lin(prefix, fin, fout, sd):
        bound = 1.0 / np.sqrt(fin)
        sd[prefix + '.weight'] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        sd[prefix + '.bias'] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    def bn(prefix, f, sd):
        sd[prefix + '.weight'] = 1.0 + 0.1 * torch.randn(f, generator=g)
        sd[prefix + '.bias'] = 0.1 * torch.randn(f, generator=g)
        sd[prefix + '.running_mean'] = 0.1 * torch.randn(f, generator=g)
        sd[prefix + '.running_var'] = 0.5 + torch.rand(f, generator=g)
        sd[prefix + '.num_batches_tracked'] = torch.tensor(100, dtype=torch.long)

    sd = {}
    nf = 63
    bn('encoder_net.1', nf, sd)
    lin('encoder_net.2', nf, num_neurons, sd)
    bn('encoder_net.4', num_neurons, sd)
    lin('encoder_net.6', num_neurons, num_neurons, sd)
    lin('encoder_net.7', num_neurons, num_neurons, sd)
    lin('encoder_net.8.mu', num_neurons, latent, sd)
    lin('encoder_net.8.logvar', num_neurons, latent, sd)
    lin('decoder_net.0', latent, num_neurons, sd)
    lin('decoder_net.3', num_neurons, num_neurons, sd)
    lin('decoder_net.5', num_neurons, 126, sd)
    return sd


def make_gmm(seed: int = 4, num_gaussians: int = 8, dim: int = 69) -> dict:
    """Synthetic content of ``gmm_08.pkl`` (hmr/smplify/prior.py:124-131)."""
    g = torch.Generator().manual_seed(seed)
    means = 0.2 * torch.randn(num_gaussians, dim, generator=g, dtype=torch.float64)
    A = 0.1 * torch.randn(num_gaussians, dim, dim, generator=g, dtype=torch.float64)
    covars = A @ A.transpose(1, 2) + 0.5 * torch.eye(dim, dtype=torch.float64)
    w = 0.5 + torch.rand(num_gaussians, generator=g, dtype=torch.float64)
    w = w / w.sum()
    return {'means': means.numpy(), 'covars': covars.numpy(), 'weights': w.numpy()}


class SyntheticSequences:
    """Array-backed stand-in for ``nemo.multi_view_sequence.MultiViewSequence``.

    Provides exactly the fields the model reads:
    ``num_views, num_frames, IMG_D0, IMG_D1, sequences[v]['pose_2d_op'|'pose_2d_gt'|'pose']``.
    """

    def __init__(self, num_views: int, num_frames: int, seed: int = 1234,
                 img_d0: int = 1080, img_d1: int = 1920, empty_frac: float = 0.02, with_eval: bool = False):
        rng = np.random.default_rng(seed)
        self.num_views = num_views
        self.num_frames = num_frames
        self.IMG_D0 = img_d0   # height
        self.IMG_D1 = img_d1   # width
        self.sequences = []
        for _ in range(num_views):
            seq = {}
            for key in ('pose_2d_op', 'pose_2d_gt'):
                kp = np.empty((num_frames, 25, 3), dtype=np.float32)
                kp[..., 0] = rng.uniform(0, img_d1, (num_frames, 25))
                kp[..., 1] = rng.uniform(0, img_d0, (num_frames, 25))
                kp[..., 2] = rng.uniform(0, 1, (num_frames, 25))
                empty = rng.uniform(size=num_frames) < empty_frac
                kp[empty] = 0.0   # undetected person (multi_view_sequence.py:422-423)
                seq[key] = [kp[t] for t in range(num_frames)]
            pose = np.zeros((num_frames, 73), dtype=np.float32)
            pose[:, 3:72] = 0.2 * rng.standard_normal((num_frames, 69))
            pose[:, 72] = (rng.uniform(size=num_frames) > 0.1).astype(np.float32)
            seq['pose'] = [pose[t] for t in range(num_frames)]
            self.sequences.append(seq)
        if with_eval:
            self._add_eval_fields(seed)

    def _add_eval_fields(self, seed):
        """Ground-truth style fields the evaluation reads (nemo/neural_motion_model.py:522-710, :1056-1282):
        ``pose_3d_gt`` (T x (72,) axis-angle), ``trans_3d_gt`` (T x (3,)), ``vibe_joints2d`` (T x (25,3)) and
        ``framerate_multiplier`` (per view).  Drawn from a SEPARATE generator so the fit inputs above (and
        every golden recorded from them) are unchanged.  The 3-D ground truth is a smooth motion that
        stands still for the first and last eighth of the clip (a non-trivial 'dynamic range')."""
        import torch
        rng = np.random.default_rng(seed + 100003)
        T = self.num_frames
        self.framerate_multiplier = [1.0 + (v % 2) for v in range(self.num_views)]
        for seq in self.sequences:
            key = 0.35 * rng.standard_normal((4, 72)).astype(np.float32)
            u = np.clip((np.arange(T) - T / 8.0) / max(T * 0.75, 1.0), 0.0, 1.0)[:, None]      # 0 .. 1, flat ends
            pose = (key[0] + np.sin(np.pi * u) * key[1] + u * key[2] + np.sin(3 * np.pi * u) * 0.5 * key[3])
            seq['pose_3d_gt'] = [torch.tensor(pose[t].astype(np.float32)) for t in range(T)]
            trans = (0.3 * rng.standard_normal(3) + u * rng.standard_normal(3)).astype(np.float32)
            seq['trans_3d_gt'] = [torch.tensor(trans[t]) for t in range(T)]
            kp = np.empty((T, 25, 3), dtype=np.float32)
            kp[..., 0] = rng.uniform(0, self.IMG_D1, (T, 25))
            kp[..., 1] = rng.uniform(0, self.IMG_D0, (T, 25))
            kp[..., 2] = rng.uniform(0, 1, (T, 25))
            seq['vibe_joints2d'] = [kp[t] for t in range(T)]

    def get_image(self, v, t):  # rendering only; never used by the fit
        raise NotImplementedError('synthetic sequences carry no images')


def published_args(**overrides) -> SimpleNamespace:
    """Hyper-parameters of the published Baseball-Pitch run
    (run_scripts_examples/nemomocap-example.sh:3-17,23-46 over config/default-v1.yml)."""
    a = dict(load_ckpt_path='', out_dir='out', model_version=2, h_dim=1000,
             instance_code_size=5, phase_rbf_dim=100, rbf_kernel='quadratic',
             monotonic_network_n_nodes=200, phase_init='linear', lr_camera=0.1,
             lr_instance=1e-3, lr_human=1e-4, lr_phase=1e-4, opt_human='adam',
             wd_human=0.001, lr_factor=1, loss='mse_robust', label_type='op',
             batch_size=512, weight_vp_loss=10, weight_vp_z_loss=1,
             weight_gmm_loss=1, weight_instance_loss=0, weight_3d_loss=0,
             code_noise=0, label_intersection_threshold=50,
             n_steps=2000, warmup_step=300, opt_cam_step=1000)
    a.update(overrides)
    return SimpleNamespace(**a)


def default_v1_args(**overrides) -> SimpleNamespace:
    """config/default-v1.yml:1-25 (NemoV1, h=500, C=10, batch 128, plateau schedulers)."""
    a = dict(load_ckpt_path='', out_dir='out', model_version=1, h_dim=500,
             instance_code_size=10, phase_rbf_dim=0, rbf_kernel='quadratic',
             monotonic_network_n_nodes=200, phase_init='linear', lr_camera=0.1,
             lr_instance=1e-3, lr_human=0.01, lr_phase=1e-5, opt_human='adam',
             wd_human=0.001, lr_factor=0.5, loss='mse_robust', label_type='op',
             batch_size=128, weight_vp_loss=0, weight_vp_z_loss=0,
             weight_gmm_loss=0.5, weight_instance_loss=0, weight_3d_loss=0,
             code_noise=0, label_intersection_threshold=50,
             n_steps=5000, warmup_step=0, opt_cam_step=1000,
             lr_pose=1e-2, lr_orient=1e-2, lr_trans=1e-2)   # (NemoV0's three networks; the script's defaults, :61-65)
    a.update(overrides)
    return SimpleNamespace(**a)
