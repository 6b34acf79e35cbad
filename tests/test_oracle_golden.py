"""Pins the oracle (oracle/) against golden vectors recorded from the real reference."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from nemo_cvpr2023_amd import synthetic as syn
from oracle import ops
from oracle.model import OracleNemo

T = torch.tensor
TOL = 2e-6


def test_rot6d_to_rotmat():
    g = load_golden('fn_rot6d_to_rotmat')
    x = T(g['x']).requires_grad_(True)
    R = ops.rot6d_to_rotmat(x)
    (R * T(g['ct'])).sum().backward()
    assert rel_err(R.detach(), g['out']) < TOL
    assert rel_err(x.grad, g['grad_x']) < 1e-5


def test_rotmat_to_aa_all_branches():
    g = load_golden('fn_rotmat_to_aa')
    R = T(g['R']).requires_grad_(True)
    aa = ops.rotmat_to_aa(R)
    (aa * T(g['ct'])).sum().backward()
    assert rel_err(aa.detach(), g['out']) < TOL
    assert np.allclose(R.grad.numpy(), g['grad_R'], rtol=1e-4, atol=1e-4 * np.abs(g['grad_R']).max())
    assert np.array_equal(ops.rotmat_to_aa(torch.eye(3).unsqueeze(0)).numpy(), g['out_identity'])
    g2 = load_golden('fn_matrot2aa')
    assert rel_err(ops.rotmat_to_aa(T(g2['R']), zero_nan=False), g2['out']) < TOL


def test_rodrigues_both_forms():
    g = load_golden('fn_batch_rodrigues')
    th = T(g['theta']).requires_grad_(True)
    R = ops.batch_rodrigues(th)
    (R * T(g['ct'])).sum().backward()
    assert rel_err(R.detach(), g['out']) < TOL
    assert rel_err(th.grad, g['grad_theta']) < 1e-5
    g = load_golden('fn_lbs_rodrigues')
    assert rel_err(ops.lbs_rodrigues(T(g['theta'])), g['out']) < TOL


def test_perspective_projection():
    g = load_golden('fn_perspective_projection')
    P, R, t = (T(g[k]).requires_grad_(True) for k in ('points', 'rotation', 'translation'))
    out = ops.perspective_projection(P, R, t, T(g['focal']), T(g['center']))
    (out * T(g['ct'])).sum().backward()
    assert rel_err(out.detach(), g['out']) < TOL
    for a, k in ((P, 'grad_points'), (R, 'grad_rotation'), (t, 'grad_translation')):
        assert rel_err(a.grad, g[k]) < 1e-5


def test_smpl_forward_and_grad():
    smpl = ops.SMPLOracle(syn.make_smpl_assets(128, seed=1))
    g = load_golden('fn_smpl_rotmat')
    rot = T(g['rotmats']).requires_grad_(True)
    v, j49, j54 = smpl.forward(T(g['betas']), rot)
    ((v * T(g['ct_vertices'])).sum() + (j49 * T(g['ct_joints'])).sum()).backward()
    assert rel_err(v.detach(), g['vertices']) < 1e-5
    assert rel_err(j49.detach(), g['joints49']) < 1e-5
    assert rel_err(j54.detach(), g['joints54']) < 1e-5
    assert rel_err(rot.grad, g['grad_rotmats']) < 1e-5
    g = load_golden('fn_smpl_betas')
    v, j49, _ = smpl.forward(T(g['betas']), T(g['rotmats']))
    assert rel_err(v, g['vertices']) < 1e-5 and rel_err(j49, g['joints49']) < 1e-5
    g = load_golden('fn_smpl_aa_eval')   # eval path: pose2rot=True, zero orient/betas
    aa = torch.cat([torch.zeros(5, 3), T(g['body_pose'])], 1)
    R = ops.lbs_rodrigues(aa.reshape(-1, 3)).reshape(5, 24, 3, 3)
    v, j49, _ = smpl.forward(torch.zeros(1, 10), R)
    assert rel_err(v, g['vertices']) < 1e-5 and rel_err(j49, g['joints49']) < 1e-5


def test_gmof_and_monotonic_and_rbf():
    g = load_golden('fn_gmof')
    assert rel_err(ops.gmof(T(g['residual']), False), g['out_sq']) < TOL
    assert rel_err(ops.gmof(T(g['residual']), True), g['out_sqrt']) < TOL
    for init in ('linear', 'rand'):
        g = load_golden('fn_monotonic_' + init)
        sh, sc, x = (T(g[k]).requires_grad_(True) for k in ('shifts', 'scales', 'x'))
        y = ops.monotonic_forward(sh, sc, x)
        (y * T(g['ct'])).sum().backward()
        assert rel_err(y.detach(), g['out']) < 1e-5
        assert rel_err(sh.grad, g['grad_shifts']) < 1e-4
        assert rel_err(sc.grad, g['grad_scales']) < 1e-4
        assert rel_err(x.grad, g['grad_x']) < 1e-4
    for kern in ops.RBF_KERNELS:
        g = load_golden('fn_rbf_' + kern)
        ls, x = T(g['log_sigmas']).requires_grad_(True), T(g['x']).requires_grad_(True)
        y = ops.rbf_forward(ls, T(g['centres']), x, kern)
        (y * T(g['ct'])).sum().backward()
        assert rel_err(y.detach(), g['out']) < TOL, kern
        assert rel_err(ls.grad, g['grad_log_sigmas']) < 1e-5, kern
        assert rel_err(x.grad, g['grad_x']) < 1e-5, kern


def test_vposer_and_gmm():
    g = load_golden('fn_vposer')
    vp = ops.VPoserOracle(syn.make_vposer_state())
    pb = T(g['pose_body']).requires_grad_(True)
    mean, scale = vp.encode(pb)
    aa, R = vp.decode(mean)
    kl = ops.kl_to_std_normal(mean, scale)
    kl.backward()
    assert rel_err(mean.detach(), g['mean']) < 1e-5 and rel_err(scale.detach(), g['scale']) < 1e-5
    assert rel_err(aa.detach(), g['dec_aa']) < 1e-5 and rel_err(R.detach(), g['dec_matrot']) < 1e-5
    assert rel_err(kl.detach(), g['kl']) < 1e-6
    assert rel_err(pb.grad, g['grad_pose_body']) < 1e-5
    g = load_golden('fn_gmm_prior')
    prior = ops.GMMPriorOracle(syn.make_gmm())
    assert np.array_equal(prior.precisions.numpy(), g['precisions'])
    assert np.array_equal(prior.nll_weights.numpy(), g['nll_weights'])
    pose = T(g['pose']).requires_grad_(True)
    ll = prior(pose)
    ll.mean().backward()
    assert rel_err(ll.detach(), g['out']) < 1e-5 and rel_err(pose.grad, g['grad_pose']) < 1e-5


def test_motionnet():
    g = load_golden('fn_motionnet')
    sd = {'m.' + k.replace('__', '.'): T(v) for k, v in g.items() if '__' in k}
    rot6d, trans = ops.motionnet_forward(sd, 'm.', T(g['x']))
    assert rel_err(rot6d, g['rot6d']) < 1e-5 and rel_err(trans, g['trans']) < 1e-5
    R = ops.rot6d_to_rotmat(rot6d).view(-1, 24, 3, 3)
    assert rel_err(R, g['rotmat']) < 1e-5
    assert rel_err(ops.rotmat_to_aa(R.reshape(-1, 3, 3)).reshape(-1, 72), g['pose']) < 1e-5


# ----------------------------------------------------------------------------- trajectories
CASES = {
    'v2_small': (2, {}, 3),
    'v1_small': (1, {'lr_human': 0.01}, 0),
    'v1_fullbatch': (1, {'batch_size': -1, 'lr_factor': 1}, 0),
    'v0_small': (0, {'lr_factor': 1}, 0),
    'v3_small': (3, {'weight_instance_loss': 0.1, 'weight_3d_loss': 0.5, 'opt_human': 'adamw'}, 2),
    'v4_small': (4, {'weight_3d_loss': 0.5}, 2),
    'v2_loss_mse': (2, {'loss': 'mse', 'weight_vp_loss': 0, 'weight_vp_z_loss': 0}, 0),
    'v2_loss_rmse': (2, {'loss': 'rmse', 'weight_vp_loss': 0, 'weight_vp_z_loss': 0}, 0),
    'v2_loss_rmse_robust': (2, {'loss': 'rmse_robust', 'weight_vp_loss': 0, 'weight_vp_z_loss': 0}, 0),
    'v2_loss_mse_robust_resized': (2, {'loss': 'mse_robust_resized', 'weight_vp_loss': 0,
                                       'weight_vp_z_loss': 0}, 0),
    # the real reference over a body model with SMPL's skinning sparsity (4 non-zero weights per vertex): what the HIP mesh
    # kernel skins sparsely (tools/gen_golden.py::run_sparse_cases)
    'v2_sparse4': (2, {}, 2),
}
# non-zero skinning weights per vertex of the synthetic body model a case was recorded with (default: all 24)
SKIN_NNZ = {'v2_sparse4': 4, 'v2_6890_sparse4': 4}


def build_case(name, cls=OracleNemo, num_verts=128, **kw):
    """Re-create the configuration tools/gen_golden.py::run_model_case used."""
    version, over, n_warm = CASES.get(name, (2, {}, 0))
    g = load_golden('model_' + name)
    V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    base = syn.published_args if version >= 2 else syn.default_v1_args
    o = dict(h_dim=48, monotonic_network_n_nodes=20, batch_size=B)
    if version >= 2:
        o['phase_rbf_dim'] = 16
    o.update(over)
    args = base(**o)
    seqs = syn.SyntheticSequences(V, Tn, seed=1234)
    state = {k[len('init__'):].replace('__', '.'): v for k, v in g.items() if k.startswith('init__')}
    model = cls(version, args, seqs, syn.make_smpl_assets(num_verts, seed=1, skin_nnz=SKIN_NNZ.get(name, 24)),
                syn.make_vposer_state(), syn.make_gmm(), state=state, **kw)
    return model, g, (V, Tn, B)


def replay(model, g, name, n_cam_default=None, tol=2e-5, check_state=True, state_tol=2e-4, la_tol=None,
           robust_state=False, drift=None):
    """drift = (first_step, tol, la_tol): looser tolerances for the optimisation steps from `first_step`
    on (recorded trajectories at lr 0.01, where Adam amplifies rounding noise; see tests/test_gpu_model.py)."""
    """Replays the script order of tools/gen_golden.py::run_model_case and checks every
    recorded number."""
    version, over, n_warm = CASES.get(name, (2, {}, 0))
    V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    torch.manual_seed(2)
    la_tol = tol if la_tol is None else la_tol
    init = {k[len('init__'):].replace('__', '.'): v for k, v in g.items() if k.startswith('init__')}

    def draw():
        return torch.randint(0, V, size=(B,)), torch.randint(0, Tn, size=(B,))

    def check(tag, ld, info, tol=tol, la_tol=la_tol):
        for k in ('gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss'):
            if f'{tag}__{k}' in g:            # (NemoV0 reports kp_loss, gmm_loss and total_loss only, :3325-3340)
                assert rel_err(ld[k], g[f'{tag}__{k}']) < tol, (tag, k, ld[k], g[f'{tag}__{k}'])
            else:
                assert k not in ld, k
        assert rel_err(ld['kp_loss'], g[f'{tag}__kp_loss_pure']) < tol, tag
        assert rel_err(info['loss_all'], g[f'{tag}__loss_all']) < la_tol, tag
        for k in ('instance_loss', 'loss_3d'):
            if f'{tag}__{k}' in g:
                assert rel_err(ld[k], g[f'{tag}__{k}']) < tol, (tag, k)

    vi, fi = draw()
    check('evalinit', *model.step(vi, fi, update=False, full_batch=True))
    vi, fi = draw()
    vi, fi = torch.as_tensor(g['preds__view_idx']), torch.as_tensor(g['preds__frame_idx'])
    with torch.no_grad():
        pd = model.get_preds_batch(vi, fi)
        p2d = model.learned_camera_projection(pd['j'], vi)
    for k in ('v', 'j', 'poses', 'orient', 'orient_aa', 'trans'):
        assert rel_err(pd[k], g['preds__' + k]) < tol, k
    assert rel_err(p2d, g['preds__points2d']) < tol
    if 'warmup_losses' in g:
        wl = model.warmup(len(g['warmup_losses']))
        assert rel_err(wl, g['warmup_losses']) < 1e-4
    if 'cam_losses' in g or 'aftercam__learned_cameras' in g:
        n_cam = len(g['cam_losses']) if len(g['cam_losses']) else n_cam_default
        cl = model.opt_cam(n_cam)
        if len(g['cam_losses']):
            assert rel_err(cl, g['cam_losses']) < 1e-4
        assert rel_err(model.state_dict()['learned_cameras'], g['aftercam__learned_cameras']) < 1e-4
    n_steps = g['batches_view'].shape[0]
    n_full = 1 if name in ('v2_small', 'v0_small', 'v2_sparse4') else 0
    for s in range(n_steps):
        draw()
        vi, fi = torch.as_tensor(g['batches_view'][s]), torch.as_tensor(g['batches_frame'][s])
        ld, info = model.step(vi, fi, full_batch=s >= n_steps - n_full)
        if drift is not None and s >= drift[0]:
            check(f'step{s}', ld, info, tol=drift[1], la_tol=drift[2])
        else:
            check(f'step{s}', ld, info)
    if check_state:
        sd = model.state_dict()
        for k, v in g.items():
            if k.startswith('final__') and not k.startswith('final__opt'):
                name_ = k[len('final__'):].replace('__', '.')
                if not robust_state:
                    assert rel_err(sd[name_], v) < state_tol, name_
                    continue
                # Adam divides by sqrt(v)+eps: an element whose gradient is rounding noise (e.g. the
                # radial direction of a rot6d column, exactly invariant) moves by an implementation-
                # dependent amount.  Compare the elements that really moved, bound the rest.
                a = sd[name_].detach().cpu().numpy().astype(np.float64)
                move = np.abs(v.astype(np.float64) - init[name_])
                if move.max() == 0:
                    assert np.abs(a - v).max() <= 1e-6 * max(np.abs(v).max(), 1e-30), name_
                    continue
                big = move >= 0.5 * move.max()
                agree = np.abs(a - v)[big] <= state_tol * np.abs(v).max() + 0.05 * move.max()
                # the rot6d head carries the exactly-invariant (noise-gradient) directions
                need = 0.75 if 'rot_out' in name_ else 0.97
                if drift is None:       # (a drifting trajectory only has to stay within twice the reference's moves)
                    assert agree.mean() >= need, (name_, float(agree.mean()))
                assert np.abs(a - v).max() <= 2.0 * move.max() + 1e-6 * np.abs(v).max(), name_
        for oi, opt in enumerate(model.optimizers):
            assert abs(opt.param_groups[0]['lr'] / float(g[f'final__opt{oi}__lr']) - 1) < 1e-6


@pytest.mark.parametrize('name', list(CASES))
def test_trajectory(name):
    model, g, _ = build_case(name)
    # NemoV0 runs three networks at lr 1e-2 on a one-dimensional input: dead ReLU units give many exactly-noise gradients,
    # which Adam turns into +-lr moves (step-0 gradients equal the reference's to the last bit, every loss of the five
    # steps to 2e-5): its final state is only held to the bound the drifting GPU replays use
    v0 = name == 'v0_small'       # (drift with a first step beyond the run: only switches the final-state check to the bound)
    replay(model, g, name, n_cam_default=3, robust_state=v0, state_tol=5e-3 if v0 else 2e-4,
           drift=(10 ** 9, 2e-5, 2e-5) if v0 else None)


def test_step0_gradients():
    model, g, _ = build_case('v2_small')
    vi, fi = torch.as_tensor(g['batches_view'][0]), torch.as_tensor(g['batches_frame'][0])
    # gradients were recorded after warmup+opt_cam; replay those first
    torch.manual_seed(2)
    V, Tn, B = 4, 7, 8
    for _ in range(2):
        torch.randint(0, V, size=(B,)), torch.randint(0, Tn, size=(B,))
    model.warmup(3)
    model.opt_cam(3)
    model.step(vi, fi)
    for k, v in g.items():
        if k.startswith('step0grad__'):
            name = k[len('step0grad__'):].replace('__', '.')
            got = model.P[name].grad
            assert rel_err(got, v) < 2e-4 or np.abs(v).max() < 1e-12, name


@pytest.mark.parametrize('name', ['v2_6890', 'v2_6890_sparse4'])
def test_full_mesh_6890_step(name):
    model, g, _ = build_case(name, num_verts=6890)
    # Adam divides by sqrt(v)+eps: entries whose gradient is a near-cancelling 20670-term sum
    # of magnitude ~eps amplify summation-order noise, so post-update weights are compared
    # loosely; the loss trajectory (the parity gate) is compared at 2e-5.
    replay(model, g, name, state_tol=2e-2)


# ------------------------------------------------------------------------------------------ next rows (8f)
def _build_eval_case():
    g = load_golden('eval_eval_v2')
    V, T = int(g['meta__V']), int(g['meta__T'])
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=8, out_dir='', phase_rbf_dim=16)
    args.model_version = 2
    seqs = syn.SyntheticSequences(V, T, seed=1234, with_eval=True)
    state = {k[len('init__'):].replace('__', '.'): torch.tensor(v) for k, v in g.items() if k.startswith('init__')}
    return g, args, seqs, state


def test_eval_metrics_match_reference_csvs():
    """oracle/evalmetrics.py == the eval_2d.csv / eval_3d.csv / eval_3d_dynamic.csv the real reference wrote."""
    from oracle import evalmetrics as em
    g, args, seqs, state = _build_eval_case()
    o = OracleNemo(2, args, seqs, syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm(),
                   state=state)
    got = em.eval_2d(o, seqs)
    for k, v in got.items():
        assert rel_err(np.asarray(v), g['eval_2d__' + k]) < 1e-4, k
    got = em.eval_3d(o, seqs)
    for k, v in got.items():
        assert rel_err(np.asarray(v), g['eval_3d__' + k]) < 1e-4, k
    got = em.eval_3d(o, seqs, dynamic_only=True)
    for k, v in got.items():
        assert rel_err(np.asarray(v), g['eval_3d_dynamic__' + k]) < 1e-4, k
    assert not np.allclose(g['eval_3d_dynamic__mpjpe-ours'], g['eval_3d__mpjpe-ours'])     # the mask is non-trivial


def test_fit_driver_reproduces_script_order():
    """nemo_cvpr2023_amd/fit.py::run_fit (duck-typed over the CPU oracle) == the loss curve the real
    reference produces when driven in the order of scripts/learned_multi_view_recon_nn.py:211-308."""
    from nemo_cvpr2023_amd.fit import run_fit
    g = load_golden('script_script_v2')
    V, T, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16,
                              n_steps=len(g['total_loss']), warmup_step=int(g['meta__n_warm']),
                              opt_cam_step=int(g['meta__n_cam']))
    args.model_version = 2
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    state = {k[len('init__'):].replace('__', '.'): torch.tensor(v) for k, v in g.items() if k.startswith('init__')}
    o = OracleNemo(2, args, seqs, syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm(),
                   state=state)
    o.num_views, o.num_frames = V, T
    torch.manual_seed(2)
    res = run_fit(o, args)
    assert rel_err(np.float32(res['init']['total_loss']), g['init_total_loss']) < 2e-5
    assert rel_err(np.asarray(res['warmup_losses']), g['warmup_losses']) < 1e-4
    assert rel_err(np.asarray(res['cam_losses']), g['cam_losses']) < 1e-4
    assert rel_err(np.asarray(res['losses']['total_loss']), g['total_loss']) < 1e-4
    assert rel_err(np.asarray([float(x['total_loss']) for x in res['evals'].values()]), g['eval_total_loss']) < 1e-4
    lrs = np.stack([res['learning_rates'][k] for k in ('lr_cam', 'lr_pose', 'lr_orient', 'lr_trans')][:g['lrs'].shape[1]], 1)
    assert np.allclose(lrs, g['lrs'][:, :lrs.shape[1]])


def test_data_layer_matches_reference_loader():
    """nemo_cvpr2023_amd/multi_view_sequence.py::load_nemo_mocap on tests/golden/mocap_fixture == the arrays
    the reference's MultiViewSequence produced from the same files (frame resampling, VIBE person selection,
    scattering of partial tracks, empty OpenPose frames, image sizes, frame-rate multipliers)."""
    import json
    from nemo_cvpr2023_amd.multi_view_sequence import load_nemo_mocap, ArrayMultiViewSequence
    g = load_golden('loader_mocap')
    root = os.path.join(os.path.dirname(__file__), 'golden', 'mocap_fixture')
    cfg = json.load(open(os.path.join(root, 'cfg.json')))
    cfg['exp_dir'] = os.path.join(root, cfg['exp_dir'])
    seqs = load_nemo_mocap(cfg, 0, 1000000, mocap_root=os.path.join(root, 'data', 'mocap'))
    assert (seqs.num_views, seqs.num_frames, seqs.IMG_D0, seqs.IMG_D1) == tuple(
        int(g[k]) for k in ('num_views', 'num_frames', 'IMG_D0', 'IMG_D1'))
    assert np.allclose(seqs.framerate_multiplier, g['framerate_multiplier'])
    for key in ('pose_2d_op', 'pose_2d_gt', 'pose', 'vibe_mask', 'vibe_joints2d', 'pose_3d_gt', 'trans_3d_gt'):
        got = np.stack([np.stack([np.asarray(x, dtype=np.float64) for x in s[key]]) for s in seqs.sequences])
        assert got.shape == g[key].shape and np.array_equal(got, g[key]), key
    assert float(np.abs(g['pose_2d_op'][1]).sum(axis=(1, 2)).min()) == 0.0      # the undetected frame is in the clip
    # the array constructor gives the same duck type
    a = ArrayMultiViewSequence.from_arrays(g['pose_2d_op'], g['pose'], seqs.IMG_D0, seqs.IMG_D1,
                                           pose_2d_gt=g['pose_2d_gt'])
    assert a.num_views == seqs.num_views and a.num_frames == seqs.num_frames
    assert np.array_equal(np.asarray(a.sequences[1]['pose'][2]), g['pose'][1, 2])


def _ckpt_case():
    g = load_golden('ckpt_ref_v2')
    V, T, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16)
    args.model_version = 2
    return g, V, T, B, args, syn.SyntheticSequences(V, T, seed=1234)


def check_resumed_run(step, g, tol=1e-4):
    """The losses the REFERENCE produced after load()-ing its own checkpoint, replayed through `step`."""
    for s in range(len(g['total_loss'])):
        ld, _ = step(torch.as_tensor(g['batches_view'][s]), torch.as_tensor(g['batches_frame'][s]),
                     full_batch=(s == 3))
        assert rel_err(ld['total_loss'], g['total_loss'][s]) < tol, (s, ld['total_loss'], g['total_loss'][s])
        assert rel_err(ld['kp_loss'], g['kp_loss_pure'][s]) < tol, s
        for k in ('gmm_loss', 'vp_recon_loss', 'vp_kl_loss'):
            assert rel_err(ld[k], g[k][s]) < tol, (s, k)


def test_checkpoint_written_by_the_reference_resumes_in_the_oracle():
    """tests/golden/ckpt_ref_v2.pt was written by the reference's own save() (tools/gen_golden.py::
    run_checkpoint_case): key names, optimiser-state layout and step counts as the reference writes them."""
    g, V, T, B, args, seqs = _ckpt_case()
    ck = torch.load(os.path.join(GOLDEN, 'ckpt_ref_v2.pt'), weights_only=False)
    torch.manual_seed(123)
    o = OracleNemo(2, args, seqs, syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm())
    assert set(ck['model_sd']) == set(o.state_dict())          # the reference's key set, nothing missing / extra
    o.load_state(ck['model_sd'])
    for opt, sd in zip(o.optimizers, ck['opt_sd']):
        opt.load_state_dict(sd)
    check_resumed_run(o.step, g)
    for k, v in o.state_dict().items():
        if k != 'phase_rbf.centres':
            assert rel_err(v, g['final__' + k.replace('.', '__')]) < 5e-3, k


def test_joints3d_smooth_loss_against_the_reference_method():
    """f-4 (BASELINE configs[4]): oracle.ops.joints3d_smooth_loss against values and gradients of the reference's own
    `FittingLoss.joints3d_smooth_loss` (humor/humor/fitting/fitting_loss.py:366-370; tools/gen_golden.py::run_smooth_case)."""
    g = load_golden('fn_joints3d_smooth_loss')
    for tag in 'abc':
        j = T(g[f'{tag}_joints']).requires_grad_(True)
        loss = ops.joints3d_smooth_loss(j)
        loss.backward()
        assert rel_err(loss.detach(), g[f'{tag}_loss']) < TOL
        assert rel_err(j.grad, g[f'{tag}_grad']) < TOL
