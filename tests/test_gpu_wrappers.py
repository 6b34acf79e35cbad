"""The reference's operator-level signatures (nemo_cvpr2023_amd/ops.py) on the HIP path against the
fixtures the REAL reference functions produced (tests/golden/fn_*.npz, tools/gen_golden.py): the same calls a reference
user would write -- rot6d_to_rotmat(x), batch_rodrigues(theta), rotation_matrix_to_angle_axis(R),
perspective_projection(points, rotation, translation, focal_length, camera_center), SMPL(...)(betas=, body_pose=,
global_orient=, pose2rot=), MaxMixturePrior(...)(pose, betas), VPoser.encode / .decode, GMoF(rho)(residual, sqrt)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from nemo_cvpr2023_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 2e-5
D = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda:0')


def test_rotation_operators_values_and_gradients():
    from nemo_cvpr2023_amd import ops
    g = load_golden('fn_rot6d_to_rotmat')
    x = D(g['x']).requires_grad_(True)
    R = ops.rot6d_to_rotmat(x)
    assert R.shape == (64, 3, 3) and rel_err(R, g['out']) < TOL
    (R * D(g['ct'])).sum().backward()
    assert rel_err(x.grad, g['grad_x']) < 1e-4
    assert rel_err(ops.rot6d_to_rotmat(D(g['x']).reshape(16, 24)), g['out']) < TOL       # anything viewable as (-1, 3, 2)

    g = load_golden('fn_batch_rodrigues')
    th = D(g['theta']).requires_grad_(True)
    R = ops.batch_rodrigues(th)
    assert rel_err(R, g['out']) < TOL
    (R * D(g['ct'])).sum().backward()
    assert rel_err(th.grad, g['grad_theta']) < 1e-4

    g = load_golden('fn_rotmat_to_aa')
    assert rel_err(ops.rotation_matrix_to_angle_axis(D(g['R'])), g['out']) < TOL
    eye = torch.eye(3, device='cuda:0').unsqueeze(0)
    assert rel_err(ops.rotation_matrix_to_angle_axis(eye), g['out_identity']) < 1e-6 or \
        float(ops.rotation_matrix_to_angle_axis(eye).abs().max()) == 0.0                   # NaN -> 0 at the identity
    R34 = torch.cat([D(g['R']), torch.zeros(128, 3, 1, device='cuda:0')], 2)              # (N, 3, 4) input, as tgm pads it
    assert rel_err(ops.rotation_matrix_to_angle_axis(R34), g['out']) < TOL


def test_perspective_projection_with_per_sample_intrinsics():
    from nemo_cvpr2023_amd import ops
    g = load_golden('fn_perspective_projection')
    out = ops.perspective_projection(D(g['points']), D(g['rotation']), D(g['translation']), D(g['focal']), D(g['center']))
    assert out.shape == (6, 25, 2) and rel_err(out, g['out']) < TOL
    # scalar focal length / one centre for the whole batch (what NeMo itself passes)
    f0, c0 = float(g['focal'][0]), g['center'][:1]
    one = ops.perspective_projection(D(g['points'][:1]), D(g['rotation'][:1]), D(g['translation'][:1]), f0, D(c0))
    assert rel_err(one, g['out'][:1]) < TOL


def test_smpl_wrapper_rotmat_betas_and_axis_angle_paths():
    from nemo_cvpr2023_amd import ops
    smpl = ops.SMPL(syn.make_smpl_assets(128, seed=1))
    g = load_golden('fn_smpl_rotmat')
    rot = D(g['rotmats'])
    out = smpl(betas=D(g['betas']), body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False)
    assert out.vertices.shape == (5, 128, 3) and out.joints.shape == (5, 49, 3)
    assert rel_err(out.vertices, g['vertices']) < TOL and rel_err(out.joints, g['joints49']) < TOL
    g = load_golden('fn_smpl_betas')                                           # non-zero betas
    rot = D(g['rotmats'])
    out = smpl(betas=D(g['betas']), body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False)
    assert rel_err(out.vertices, g['vertices']) < TOL and rel_err(out.joints, g['joints49']) < TOL
    g = load_golden('fn_smpl_aa_eval')                                         # the evaluation call: aa pose, no betas / orient
    out = smpl(betas=None, body_pose=D(g['body_pose']), global_orient=None, pose2rot=True)
    assert rel_err(out.vertices, g['vertices']) < TOL and rel_err(out.joints, g['joints49']) < TOL


def test_gmof_prior_and_vposer_wrappers():
    from nemo_cvpr2023_amd import ops
    g = load_golden('fn_gmof')
    rob = ops.GMoF(rho=100)
    assert rel_err(rob(D(g['residual']), False), g['out_sq']) < TOL and rel_err(rob(D(g['residual']), True), g['out_sqrt']) < TOL

    g = load_golden('fn_gmm_prior')
    prior = ops.MaxMixturePrior(syn.make_gmm())
    pose = D(g['pose']).requires_grad_(True)
    ll = prior(pose, None)
    assert ll.shape == (10,) and rel_err(ll, g['out']) < TOL
    ll.mean().backward()
    assert rel_err(pose.grad, g['grad_pose']) < 1e-4

    g = load_golden('fn_vposer')
    vp = ops.VPoser(syn.make_vposer_state())
    q = vp.encode(D(g['pose_body']))
    assert rel_err(q.mean, g['mean']) < TOL and rel_err(q.scale, g['scale']) < TOL
    dec = vp.decode(q.mean)
    assert dec['pose_body'].shape == (6, 21, 3) and dec['pose_body_matrot'].shape == (6, 21, 9)
    assert rel_err(dec['pose_body'], g['dec_aa']) < TOL and rel_err(dec['pose_body_matrot'], g['dec_matrot']) < TOL
