"""GPU parity of the drop-in NemoV* classes: the trajectories recorded from the real reference
(tests/golden/model_*.npz) replayed through the HIP engine, plus full-size checks against the oracle."""
import numpy as np
import os
import pytest
import torch

from conftest import load_golden, rel_err
from nemo_cvpr2023_amd import synthetic as syn
from test_oracle_golden import CASES, SKIN_NNZ, replay

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build_hip_case(name, num_verts=128, tmp_path=None):
    from nemo_cvpr2023_amd.neural_motion_model import NEMO_VERSIONS
    version, over, _ = CASES.get(name, (2, {}, 0))
    g = load_golden('model_' + name)
    V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    base = syn.published_args if version >= 2 else syn.default_v1_args
    o = dict(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir=str(tmp_path) if tmp_path else '')
    if version >= 2:
        o['phase_rbf_dim'] = 16
    o.update(over)
    args = base(**o)
    seqs = syn.SyntheticSequences(V, Tn, seed=1234)
    torch.manual_seed(0)
    m = NEMO_VERSIONS[version](args, seqs, DEV, smpl_assets=syn.make_smpl_assets(num_verts, seed=1, skin_nnz=SKIN_NNZ.get(name, 24)),
                               vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    state = {k[len('init__'):].replace('__', '.'): torch.tensor(v) for k, v in g.items()
             if k.startswith('init__')}
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return m, g


# Per-step parity on identical state is ~1e-5 (see test_lockstep_with_oracle); over a recorded
# trajectory Adam's scale-free update amplifies rounding-level gradient differences, fastest for the
# default-v1 cases (lr_human = 0.01, 100x the published run).  Scalar losses stay within `tol`,
# individual per-joint losses within `la_tol`.
# v1_small (30 recorded steps at lr 0.01): step 0 agrees to 6e-7 / 6e-6; from step 1 on Adam has moved every
# noise-level-gradient weight by +-lr according to the SIGN of rounding noise, so the recorded drift depends
# on the fp32 summation order inside the kernels (tools/debug_traj.py prints the per-step table: loss terms
# up to 1.4e-2, single per-joint losses up to 8e-2 over the 30 steps).  The first step is held to the tight
# bound, the rest of that trajectory only to a sanity bound; per-step parity is test_lockstep_with_oracle's job.
TRAJ_TOL = {'v1_small': (1e-4, 1e-4), 'v1_fullbatch': (1e-4, 1e-3)}
TRAJ_DRIFT = {'v1_small': (1, 5e-2, 5e-1),
              # NemoV0 (three networks at lr 1e-2 on a 1-D input): the same regime -- step 0 at 1e-4, later steps to the
              # drift bound; per-step parity is test_lockstep_with_oracle['v0_small']
              'v0_small': (1, 5e-2, 5e-1)}


@pytest.mark.parametrize('name', list(CASES))
def test_reference_trajectory(name, tmp_path):
    m, g = build_hip_case(name, tmp_path=tmp_path)
    tol, la_tol = TRAJ_TOL.get(name, (1e-4, 1e-4))
    replay(m, g, name, n_cam_default=3, tol=tol, la_tol=la_tol, state_tol=5e-3, robust_state=True,
           drift=TRAJ_DRIFT.get(name))


def test_v1_small_every_recorded_step_from_the_reference_state(tmp_path):
    """The default-v1 trajectory (30 recorded steps at lr_human = 0.01, plateau schedulers on) step by step: the CPU
    oracle replays the recording (it tracks the reference to 2e-5 on every number, tests/test_oracle_golden.py), and
    before EVERY step the HIP model takes over the oracle's parameters, Adam moments, step counts, learning rates
    and scheduler state -- i.e. (to 2e-5) the reference's own state at that step.  From there the HIP step must
    reproduce the numbers the REFERENCE recorded for that step to 1e-4: every loss term and every per-joint loss of
    all 30 steps, not only the first one (the free-running replay above can only hold steps >= 1 to a drift bound,
    because Adam turns rounding noise on noise-level gradients into +-lr moves)."""
    from test_oracle_golden import build_case
    name = 'v1_small'
    m, g = build_hip_case(name, tmp_path=tmp_path)
    o, _, (V, Tn, B) = build_case(name)
    torch.manual_seed(2)
    o.opt_cam(2)                                        # (eval-at-init / get_preds draw nothing from the model state)
    n_steps = g['batches_view'].shape[0]
    assert n_steps == 30 and len(m.schedulers) == len(o.schedulers) > 0
    for s in range(n_steps):
        m.load_state_dict({k: v for k, v in o.state_dict().items()}, strict=False)
        for mo, oo in zip(m.optimizers, o.optimizers):
            sd = oo.state_dict()
            if sd['state']:
                mo.load_state_dict(sd)
            mo.param_groups[0]['lr'] = oo.param_groups[0]['lr']
        for ms, os_ in zip(m.schedulers, o.schedulers):
            ms.load_state_dict({k: v for k, v in os_.state_dict().items() if k != 'optimizer'})
        vi, fi = torch.as_tensor(g['batches_view'][s]), torch.as_tensor(g['batches_frame'][s])
        ld_h, info_h = m.step(vi, fi)
        ld_o, _ = o.step(vi, fi)
        tag = f'step{s}'
        for k in ('gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss'):
            assert rel_err(ld_h[k], g[f'{tag}__{k}']) < 1e-4, (s, k, ld_h[k], g[f'{tag}__{k}'])
            assert rel_err(ld_o[k], g[f'{tag}__{k}']) < 2e-5, (s, k)          # the oracle is still on the recording
        assert rel_err(ld_h['kp_loss'], g[f'{tag}__kp_loss_pure']) < 1e-4, s
        assert rel_err(info_h['loss_all'], g[f'{tag}__loss_all']) < 1e-4, s
    lrs = [o_.param_groups[0]['lr'] for o_ in o.optimizers]
    for i, lr in enumerate(lrs):                         # (the plateau schedulers dropped the rates on the way)
        assert abs(lr / float(g[f'final__opt{i}__lr']) - 1) < 1e-6


@pytest.mark.parametrize('name', ['v2_6890', 'v2_6890_sparse4'])
def test_reference_trajectory_full_mesh(name, tmp_path):
    """The real reference's recorded steps over the 6890-vertex mesh -- with dense skinning weights and with SMPL's four
    non-zero weights per vertex, which the HIP mesh kernel skins sparsely."""
    m, g = build_hip_case(name, num_verts=6890, tmp_path=tmp_path)
    assert m.engine.ctx.skin_sparse == name.endswith('sparse4')
    replay(m, g, name, tol=1e-4, state_tol=2e-2, robust_state=True)


def _float64_twin(o):
    """Deep copy of an oracle model with every tensor in float64: the arbiter for ill-conditioned
    quantities (e.g. saturated-sigmoid derivatives in the phase networks, 1 - sigma(15) in fp32)."""
    import copy
    t = copy.deepcopy(o)
    for k in t.P:
        t.P[k] = t.P[k].detach().double().requires_grad_(True)
    for attr in ('points2d_gt_all', 'gt_bbox_size', 'hmr_theta', 'hmr_mask', 'rbf_centres'):
        if hasattr(t, attr):
            setattr(t, attr, getattr(t, attr).double())
    for k, v in t.smpl.a.items():
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            t.smpl.a[k] = v.double()
    t.vp.sd = {k: v.double() for k, v in t.vp.sd.items()}
    t.prior.means, t.prior.precisions = t.prior.means.double(), t.prior.precisions.double()
    t.prior.nll_weights = t.prior.nll_weights.double()
    t._build_optimizers()
    return t


@pytest.mark.parametrize('name', ['v0_small', 'v1_small', 'v2_small', 'v3_small', 'v4_small'])
def test_lockstep_with_oracle(name, tmp_path):
    """Strict single-step parity over a whole run: before every step the oracle is re-synchronised
    to the HIP model's parameters AND Adam state (through the torch-format state_dicts), then both
    take the same step.  Losses / joints must agree to 2e-5; every gradient must be as close to the
    float64 truth as the reference's own fp32 arithmetic is (x3) or within 5e-4 of the optimiser
    group's gradient scale; updated parameters are compared where the gradient is well-conditioned."""
    from test_oracle_golden import build_case
    m, g = build_hip_case(name, tmp_path=tmp_path)
    o, _, _ = build_case(name)
    V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    torch.manual_seed(3)
    named = dict(m.named_parameters())
    for s in range(8):
        o.load_state({k: v.cpu() for k, v in m.state_dict().items()})
        for oo, mo in zip(o.optimizers, m.optimizers):
            sd = mo.state_dict()
            if sd['state']:
                oo.load_state_dict(sd)
        vi, fi = torch.randint(0, V, (B,)), torch.randint(0, Tn, (B,))
        o64 = _float64_twin(o)
        torch.set_default_dtype(torch.float64)
        try:
            o64.step(vi, fi)
        finally:
            torch.set_default_dtype(torch.float32)
        ld_h, info_h = m.step(vi, fi)
        ld_o, info_o = o.step(vi, fi)
        for k in ld_o:
            assert rel_err(ld_h[k], ld_o[k]) < 2e-5 or abs(float(ld_o[k])) < 1e-12, (s, k)
        assert rel_err(info_h['loss_all'], info_o['loss_all']) < 5e-5, s
        assert rel_err(info_h['j'], info_o['j']) < 2e-5, s
        scale = {}
        for oo in o.optimizers:
            gmax = max([float(q.grad.abs().max()) for q in oo.param_groups[0]['params'] if q.grad is not None]
                       + [0.0])
            for q in oo.param_groups[0]['params']:
                scale[id(q)] = gmax
        sd_h = m.state_dict()
        for k, p in o.P.items():
            if k == 'learned_betas' or p.grad is None:
                continue
            g64 = o64.P[k].grad
            e_h = (named[k].grad.cpu().double() - g64).abs()
            e_o = (p.grad.double() - g64).abs()
            assert float(e_h.max()) <= 5e-4 * scale[id(p)] + 3.0 * float(e_o.max()), (s, k, float(e_h.max()))
            # Adam update where the gradient is far above its own fp32 uncertainty
            well = (g64.abs() > 1e-3 * float(g64.abs().max())) & (e_o < 1e-4 * g64.abs())
            if well.any():
                d = (sd_h[k].cpu() - p.detach()).abs()[well].max()
                lr = [oo.param_groups[0]['lr'] for oo in o.optimizers
                      if any(q is p for q in oo.param_groups[0]['params'])][0]
                assert float(d) < 0.02 * lr + 1e-6 * float(p.detach().abs().max()), (s, k, float(d))


def test_step0_gradients_vs_reference(tmp_path):
    m, g = build_hip_case('v2_small', tmp_path=tmp_path)
    torch.manual_seed(2)
    V, Tn, B = 4, 7, 8
    for _ in range(2):
        torch.randint(0, V, size=(B,)), torch.randint(0, Tn, size=(B,))
    m.warmup(3)
    m.opt_cam(3)
    m.step(torch.as_tensor(g['batches_view'][0]), torch.as_tensor(g['batches_frame'][0]))
    named = dict(m.named_parameters())
    checked = 0
    for k, v in g.items():
        if k.startswith('step0grad__'):
            name = k[len('step0grad__'):].replace('__', '.')
            if name == 'learned_betas':
                continue           # in no optimiser; its gradient is deliberately not computed
            if np.abs(v).max() < 1e-12:
                assert float(named[name].grad.abs().max()) == 0.0, name
            else:
                assert rel_err(named[name].grad, v) < 1e-3, name
            checked += 1
    assert checked >= 20


def _assert_gradients_up_to_l1_ties(m, o, o64, bound, keys):
    """|g_hip - g_oracle| <= 1e-4 of the tensor's largest entry + 3 x the oracle's own fp32-vs-float64 distance (as in
    test_every_gradient_at_real_size_with_the_mesh_term_off) + what sign(0) ties of the L1 mesh term can change (tiebound.py:
    elementwise, from the float64 twin) -- the flat 2e-3 of rounds 1 - 4 is gone."""
    named = dict(m.named_parameters())
    worst = 0.0
    for k in keys:
        gh, go, g64 = named[k].grad.detach().cpu().double(), o.P[k].grad.double(), o64.P[k].grad.double()
        scale = float(go.abs().max())
        noise = float((go - g64).abs().max())
        allow = 1e-4 * scale + 3.0 * noise + bound[k].double()
        diff = (gh - go).abs()
        if scale > 0:         # (a gradient that is exactly zero on the oracle's side -- the cancelled linear_out bias -- must be zero here too)
            worst = max(worst, float((diff - 3.0 * noise - bound[k].double()).max()) / scale)
        assert bool((diff <= allow).all()), (k, float(diff.max()), scale, noise, float(bound[k].max()))
    return worst


def test_published_config_step_vs_oracle(tmp_path):
    """One real-size step (NemoV2, h=1000, RBF 100, 6890 vertices, minibatch 512 drawn from 8x300)
    against the CPU oracle on the same initial state: losses, 3-D joints, 2-D points."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T, B = 8, 300, 512
    args = syn.published_args(batch_size=B, out_dir=str(tmp_path))
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    # leave the near-identity regime so that every term is exercised
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state=state)
    torch.manual_seed(2)
    vi, fi = torch.randint(0, V, (B,)), torch.randint(0, T, (B,))
    from tiebound import model_v2v_tie_bound
    o64 = _float64_twin(o)
    bound, n_ties = model_v2v_tie_bound(o64, vi, fi)
    torch.set_default_dtype(torch.float64)
    try:
        o64.step(vi, fi, update=True)
    finally:
        torch.set_default_dtype(torch.float32)
    ld_o, info_o = o.step(vi, fi, update=True)
    ld_h, info_h = m.step(vi, fi, update=True)
    for k in ('kp_loss', 'gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss'):
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4, (k, ld_h[k], ld_o[k])
    assert rel_err(info_h['j'], info_o['j']) < 1e-4
    assert rel_err(info_h['points2d'], info_o['points2d']) < 1e-4
    assert rel_err(info_h['loss_all'], info_o['loss_all']) < 1e-4
    # gradients of the big shared tensors (before Adam they are in .grad on both sides)
    named = dict(m.named_parameters())
    worst = _assert_gradients_up_to_l1_ties(m, o, o64, bound, ('learned_motion.net.net.2.weight', 'learned_motion.rot_out.weight',
                                                               'learned_cameras', 'phase_rbf.log_sigmas', 'learned_instance_code',
                                                               'phase_networks.3.shifts'))
    print('L1 coordinates within rounding of a tie:', n_ties, ' worst gradient error beyond the allowances / scale:', worst)
    # second step: the post-update state must produce matching losses too
    vi, fi = torch.randint(0, V, (B,)), torch.randint(0, T, (B,))
    ld_o, _ = o.step(vi, fi, update=False)
    ld_h, _ = m.step(vi, fi, update=False)
    for k in ('kp_loss', 'total_loss'):
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4, k


@pytest.mark.parametrize('version,skin_nnz', [(2, 4), (2, 24), (3, 24), (4, 4)])
def test_benchmark_config_full_batch_step_vs_oracle(version, skin_nnz):
    """THE benchmark workload (8 x 300 full batch, N = 2400, h = 1000, RBF 100, 6890 vertices, every loss term; the body
    model with the published SMPL model's 4 non-zero skinning weights per vertex -- the sparse-skinning mesh kernel -- and
    with a dense weight matrix) --
    and the same sizes for NemoV3 / NemoV4 with their extra terms on: one update step and one evaluation step
    against the CPU oracle from the same state.  At this size the mesh kernel runs its 3-range + left-over-block
    grid (150 sample groups) and the hidden-layer GEMMs their whole-tiles + split-tail schedule."""
    from nemo_cvpr2023_amd.neural_motion_model import NEMO_VERSIONS
    from oracle.model import OracleNemo
    V, T = 8, 300
    args = syn.published_args(batch_size=512, out_dir='')
    if version >= 3:
        args.weight_instance_loss, args.weight_3d_loss = 0.1, 0.5
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=skin_nnz), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NEMO_VERSIONS[version](args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    assert m.engine.ctx.skin_sparse == (skin_nnz <= 4)
    with torch.no_grad():                      # leave the near-identity regime so that every term is exercised
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(version, args, seqs, assets, vps, gmm,
                   state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    from tiebound import model_v2v_tie_bound
    o64 = _float64_twin(o)
    bound, n_ties = model_v2v_tie_bound(o64, *o64.full_indices())
    torch.set_default_dtype(torch.float64)
    try:
        o64.step(None, None, update=True, full_batch=True)
    finally:
        torch.set_default_dtype(torch.float32)
    ld_o, info_o = o.step(None, None, update=True, full_batch=True)
    ld_h, info_h = m.step(None, None, update=True, full_batch=True)
    assert ld_h.keys() == ld_o.keys()
    for k in ld_o:
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4, (k, ld_h[k], ld_o[k])
    assert rel_err(info_h['loss_all'], info_o['loss_all']) < 1e-4
    named = dict(m.named_parameters())
    worst = _assert_gradients_up_to_l1_ties(m, o, o64, bound, ('learned_motion.net.net.0.weight', 'learned_motion.net.net.2.weight',
                                                               'learned_motion.rot_out.weight', 'learned_cameras',
                                                               'phase_rbf.log_sigmas', 'phase_networks.5.scales'))
    print('L1 coordinates within rounding of a tie:', n_ties, ' worst gradient error beyond the allowances / scale:', worst)
    ld_o, _ = o.step(None, None, update=False, full_batch=True)      # the updated state
    ld_h, _ = m.step(None, None, update=False, full_batch=True)
    for k in ld_o:
        assert rel_err(ld_h[k], ld_o[k]) < 2e-4, (k, ld_h[k], ld_o[k])


def _grad_table(m, o, o64=None):
    """(name, HIP-vs-oracle error, tensor scale, oracle-fp32-vs-float64 error or None) for every parameter with a gradient."""
    named = dict(m.named_parameters())
    out = []
    for k, p in o.P.items():
        if k == 'learned_betas' or p.grad is None:
            continue
        gh = named[k].grad.detach().cpu().double()
        out.append((k, float((gh - p.grad.double()).abs().max()), float(p.grad.abs().max()),
                    None if o64 is None else float((p.grad.double() - o64.P[k].grad).abs().max())))
    return out


@pytest.mark.parametrize('B', [512, 2400])
def test_every_gradient_at_real_size_with_the_mesh_term_off(B):
    """SURVEY 8(d)'s gradient bar at the REAL sizes (h = 1000, RBF 100, K = 200, 6890 vertices; a minibatch of 512 and
    the full 8 x 300 batch): with weight_vp_loss = 0 -- no L1 term, whose sign(0) ties are the one place where two
    correct fp32 evaluations may differ by more than rounding -- EVERY parameter gradient of one update step is within
    1e-4 of the oracle's (relative to the tensor's largest entry) plus the oracle's own fp32-vs-float64 distance -- which is
    what lets the ill-conditioned phase-network gradients (differences of saturated sigmoids) in --, and no shared, camera
    or code gradient is further than 2e-4 in any case."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T = 8, 300
    args = syn.published_args(batch_size=512, out_dir='', weight_vp_loss=0)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    torch.manual_seed(4)
    if B == 2400:
        call = lambda mdl: mdl.step(None, None, update=True, full_batch=True)
    else:
        vi, fi = torch.randint(0, V, (B,)), torch.randint(0, T, (B,))
        call = lambda mdl: mdl.step(vi, fi, update=True)
    o64 = _float64_twin(o)
    torch.set_default_dtype(torch.float64)
    try:
        call(o64)
    finally:
        torch.set_default_dtype(torch.float32)
    ld_o, _ = call(o)
    ld_h, _ = call(m)
    for k in ld_o:
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4 or abs(float(ld_o[k])) < 1e-12, (k, ld_h[k], ld_o[k])
    table = _grad_table(m, o, o64)
    assert len(table) >= 28
    worst = max(err / scale for k, err, scale, _ in table if not k.startswith('phase_networks.') and scale > 0)
    print('worst non-phase gradient error / scale:', worst)
    for k, err, scale, noise in table:
        # 1e-4 of the tensor's scale, plus what the ORACLE's own fp32 arithmetic is away from a float64 evaluation of the
        # same step (sums over 2400 samples: 1e-5 of the scale for the instance codes; the phase networks, differences of
        # saturated sigmoids: up to 1e-2)
        assert err <= 1e-4 * scale + 3.0 * noise, (k, err, scale, noise)
        if not k.startswith('phase_networks.'):
            assert err <= 2e-4 * scale, (k, err, scale)


def test_default_v1_at_its_real_hyper_parameters_in_lock_step():
    """config/default-v1.yml as shipped (NemoV1, h 500, K 200, C 10, batch 128, lr_human 0.01, plateau schedulers with
    factor 0.5; BASELINE configs[0]: 1 instance x 30 frames) on the 6890-vertex mesh: 30 update steps in lock step -- the
    CPU oracle drives, before every step the HIP model takes over its parameters, Adam state, learning rates and
    scheduler state; every loss term and per-joint loss of every step within 1e-4, every gradient within 1e-3 of its
    scale (the V = 1 branch of engine.ldp and the single-view keypoint mean at real sizes)."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV1
    from oracle.model import OracleNemo
    V, T = 1, 30
    args = syn.default_v1_args(out_dir='')
    assert (args.h_dim, args.monotonic_network_n_nodes, args.instance_code_size, args.batch_size, args.lr_factor) == \
        (500, 200, 10, 128, 0.5)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV1(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    assert m.engine.ldp == 2 * 200 + 8 and len(m.schedulers) == 4
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(1, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(6)
    first_lr = o.optimizers[1].param_groups[0]['lr']
    for s in range(30):
        m.load_state_dict({k: v for k, v in o.state_dict().items()}, strict=False)
        for mo, oo in zip(m.optimizers, o.optimizers):
            sd = oo.state_dict()
            if sd['state']:
                mo.load_state_dict(sd)
            mo.param_groups[0]['lr'] = oo.param_groups[0]['lr']
        for ms, os_ in zip(m.schedulers, o.schedulers):
            ms.load_state_dict({k: v for k, v in os_.state_dict().items() if k != 'optimizer'})
        vi, fi = torch.randint(0, V, (128,), generator=gen), torch.randint(0, T, (128,), generator=gen)
        ld_o, info_o = o.step(vi, fi)
        ld_h, info_h = m.step(vi, fi)
        assert ld_h.keys() == ld_o.keys()
        for k in ld_o:
            assert rel_err(ld_h[k], ld_o[k]) < 1e-4 or abs(float(ld_o[k])) < 1e-12, (s, k, ld_h[k], ld_o[k])
        assert rel_err(info_h['loss_all'], info_o['loss_all']) < 1e-4, s
        assert rel_err(info_h['j'], info_o['j']) < 1e-4, s
        for k, err, scale, _ in _grad_table(m, o):
            if not k.startswith('phase_networks.'):
                assert err <= 1e-3 * scale, (s, k, err, scale)
    for mo, oo in zip(m.optimizers, o.optimizers):
        assert mo.param_groups[0]['lr'] == oo.param_groups[0]['lr']


@pytest.mark.parametrize('skin_nnz', [24, 4])
def test_more_than_one_mesh_chunk_vs_oracle(skin_nnz):
    """N > 8192: the full-mesh term is processed in 8192-sample chunks (so that dVP^T stays bounded at any N);
    a 4 x 2058 full batch (N = 8232: one whole chunk + a ragged 40-sample one) on a small mesh against the oracle;
    dense and SMPL-sparse skinning weights."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T = 4, 2058
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=64, out_dir='', phase_rbf_dim=16)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(100, seed=1, skin_nnz=skin_nnz), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    for it in range(3):
        ld_o, _ = o.step(None, None, update=True, full_batch=True)
        ld_h, _ = m.step(None, None, update=True, full_batch=True)
        for k in ld_o:
            assert rel_err(ld_h[k], ld_o[k]) < 1e-4, (it, k, ld_h[k], ld_o[k])
        # gradients on BOTH steps: the second one runs over scratch the first one's ragged last chunk has used (a
        # launch-size dependent scratch layout once left that chunk's data where the next step expected zeroed tickets)
        named = dict(m.named_parameters())
        for k in ('learned_motion.net.net.2.weight', 'learned_motion.rot_out.weight', 'learned_cameras'):
            assert rel_err(named[k].grad, o.P[k].grad) < (2e-3 if it == 0 else 1e-2), (it, k)


@pytest.mark.parametrize('B,one_view', [(1, False), (5, False), (17, True), (33, False)])
def test_ragged_and_degenerate_minibatches_vs_oracle(B, one_view):
    """Minibatches that are not a multiple of any tile: a single sample, 5, 17 (all from ONE view, the other
    views absent from the per-view mean), 33 with repeated (view, frame) pairs -- update steps against the oracle."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T = 3, 10
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(100, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(B)
    for it in range(3):
        vi = torch.full((B,), 1, dtype=torch.long) if one_view else torch.randint(0, V, (B,), generator=gen)
        fi = torch.randint(0, T, (B,), generator=gen)
        ld_o, info_o = o.step(vi, fi)
        ld_h, info_h = m.step(vi, fi)
        for k in ld_o:
            assert rel_err(ld_h[k], ld_o[k]) < 1e-4, (it, k, ld_h[k], ld_o[k])
        assert rel_err(info_h['loss_all'], info_o['loss_all']) < 1e-4


def _tiny_v2(seed=0, version=2, **over):
    from nemo_cvpr2023_amd.neural_motion_model import NEMO_VERSIONS
    V, T = 3, 10
    o = dict(h_dim=48, monotonic_network_n_nodes=20, batch_size=9, out_dir='', phase_rbf_dim=16)
    o.update(over)
    args = syn.published_args(**o)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(100, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(seed)
    m = NEMO_VERSIONS[version](args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    return m, (version, args, seqs, assets, vps, gmm)


def _copy_model_state(dst, src):
    dst.load_state_dict({k: v.detach().clone() for k, v in src.state_dict().items()}, strict=False)
    for do, so in zip(dst.optimizers, src.optimizers):
        sd = so.state_dict()
        if sd['state']:
            do.load_state_dict(sd)


@pytest.mark.parametrize('mode', ['step', 'warmup'])
def test_smaller_batch_after_a_larger_one_in_a_shared_workspace(mode):
    """Workspaces are shared by all batch sizes that round up to the same multiple of 64 (engine.ws_capacity), so a
    step of N samples runs over buffers an earlier step of N' > N samples has written.  Row N of the head gradient
    (the 'phase 0 / zero code' row behind trans_0) must not keep sample N's rotation gradient of the larger batch:
    model A steps 9 samples and then 7, model B (fresh engine) takes over A's state after the first step and steps
    the same 7 -- every flat gradient must agree (same kernels on the same inputs: 1e-5 leaves room for the atomics
    of the phase backward) -- and both must agree with the oracle."""
    from oracle.model import OracleNemo
    a, (version, args, seqs, assets, vps, gmm) = _tiny_v2()
    b, _ = _tiny_v2()
    with torch.no_grad():
        a.learned_motion.rot_out.weight.mul_(2e3)
    gen = torch.Generator().manual_seed(5)
    draw = lambda n: (torch.randint(0, 3, (n,), generator=gen), torch.randint(0, 10, (n,), generator=gen))
    if mode == 'step':
        a.step(*draw(9))
        _copy_model_state(b, a)
        o = OracleNemo(version, args, seqs, assets, vps, gmm,
                       state={k: v.detach().cpu() for k, v in a.state_dict().items()})
        vi, fi = draw(7)
        a.step(vi, fi)
        b.step(vi, fi)
        o.step(vi, fi)
    else:
        torch.manual_seed(1)
        a.warmup(1)                       # batch_size 9
        _copy_model_state(b, a)
        o = OracleNemo(version, args, seqs, assets, vps, gmm,
                       state={k: v.detach().cpu() for k, v in a.state_dict().items()})
        args.batch_size = b.args.batch_size = 7
        for mdl in (a, b, o):
            torch.manual_seed(2)
            mdl.warmup(1)
    ga, gb = a.engine.grads, b.engine.grads
    assert float(gb.abs().max()) > 0
    assert rel_err(ga, gb) < 1e-5, float((ga - gb).abs().max())
    named = dict(a.named_parameters())
    for k, p in o.P.items():
        if k == 'learned_betas' or p.grad is None or float(p.grad.abs().max()) == 0:
            continue
        assert rel_err(named[k].grad, p.grad) < 2e-3, k


def test_info_dict_of_a_training_step_is_private_when_the_caller_passed_device_indices():
    """with device-resident indices (what the unchanged reference script hands over, scripts:291-300) the
    tensors of a training step's info_dict are private copies, as in the reference -- they survive the next step; with
    host-drawn indices they are materialised on first access (and equal the eager ones when read in time)."""
    m, _ = _tiny_v2()
    gen = torch.Generator().manual_seed(4)
    draw = lambda: (torch.randint(0, 3, (9,), generator=gen), torch.randint(0, 10, (9,), generator=gen))
    for _ in range(3):                      # (eager, captured, replayed)
        vi, fi = draw()
        _, info = m.step(vi.to(DEV), fi.to(DEV))
        assert not getattr(info, 'lazy', None), info.lazy
        kept = {k: info[k].clone() for k in ('loss_all', 'points2d', 'j', 'points2d_gt')}
        vi2, fi2 = draw()
        _, info2 = m.step(vi2, fi2)             # host indices: lazy
        assert set(info2.lazy) == {'loss_all', 'points2d', 'j', 'points2d_gt'}
        for k, v in kept.items():
            assert torch.equal(info[k], v), k   # the earlier step's tensors were not overwritten
        assert info2['j'].shape == kept['j'].shape and not torch.equal(info2['j'], kept['j'])


@pytest.mark.parametrize('graphs', [True, False])
def test_warmup_raises_on_nan_gradients_and_keeps_the_parameters(graphs):
    """:3497-3500 ("nan gradient found" -> ipdb) on the captured warm-up: the NaN-gradient count is a device scalar, the
    update is skipped from the first poisoned iteration on, and the phase ends with FloatingPointError -- the parameters are
    the ones before that iteration (here: three clean iterations, then a NaN in the 3-D pose targets)."""
    m, _ = _tiny_v2()
    m.use_graphs = graphs
    torch.manual_seed(3)
    clean = m.warmup(3)
    assert len(clean) == 3 and all(np.isfinite(clean))
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.engine.hmr_theta[:] = float('nan')
    with pytest.raises(FloatingPointError):
        m.warmup(4)
    for k, v in m.state_dict().items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, before[k]), k


@pytest.mark.parametrize('version', [2, 3])
def test_padded_launch_equals_the_unpadded_step(version):
    """A minibatch launched with masked padding rows (what a rank of a sharded run does with its share of a random
    draw, so that one captured graph serves every share size that rounds up to the same multiple of 32) is the same
    step: losses, per-joint losses, every gradient and the updated parameters equal the unpadded launch of the same
    samples -- for batches with an absent view, for a batch that is already a multiple of the pad, and for NemoV3's
    extra terms -- and three different sizes that round up to 32 share ONE captured graph."""
    from nemo_cvpr2023_amd.neural_motion_model import ShardInfo
    over = dict(weight_instance_loss=0.1, weight_3d_loss=0.5) if version >= 3 else {}
    a, _ = _tiny_v2(version=version, **over)
    b, _ = _tiny_v2(version=version, **over)
    with torch.no_grad():
        a.learned_motion.rot_out.weight.mul_(2e3)
    _copy_model_state(b, a)
    gen = torch.Generator().manual_seed(9)
    sizes = [19, 27, 23, 19, 32, 27, 40, 5, 45]
    for it, n in enumerate(sizes):
        vi, fi = torch.randint(0, 3, (n,), generator=gen), torch.randint(0, 10, (n,), generator=gen)
        if it == 2:
            vi[vi == 1] = 2                                  # a view absent from the batch
        # (as one rank of a sharded run sees it: its n samples out of a global batch of 64 -- per-sample means are
        #  scaled by n / 64 on both sides, so both models compute the same gradients)
        ld_a, info_a = a.step(vi, fi, _shard=ShardInfo(mr=n / 64.0, n_global=64))
        ld_b, info_b = b.step(vi, fi, _shard=ShardInfo(n_global=64, pad=32))
        for k in ld_a:
            assert rel_err(ld_b[k], ld_a[k]) < 1e-5 or abs(float(ld_a[k])) < 1e-12, (it, n, k, ld_a[k], ld_b[k])
        assert info_b['loss_all'].shape == info_a['loss_all'].shape and info_b['j'].shape[0] == n
        assert rel_err(info_b['loss_all'], info_a['loss_all']) < 1e-5
        assert torch.equal(info_b['view_idx'].cpu(), vi)
        ga, gb = a.engine.grads, b.engine.grads
        named_a, named_b = dict(a.named_parameters()), dict(b.named_parameters())
        for k in named_a:
            if k == 'learned_betas':
                continue
            sc = float(named_a[k].grad.abs().max())
            assert float((named_a[k].grad - named_b[k].grad).abs().max()) <= 2e-5 * sc + 1e-12, (it, n, k)
        _copy_model_state(b, a)           # (keep the two in lock step: Adam amplifies rounding-level differences)
    keys = [k for w in b.engine.ws.values() for k, g in w['graphs'].items() if isinstance(g, torch.cuda.CUDAGraph)]
    if os.environ.get('NEMO_GRAPHS', '1') != '0':
        assert sorted(k[1] for k in keys) == [32, 64], keys      # launch sizes 32 (19, 27, 23, 32, 5 samples) and 64 (40, 45)
    assert b.launch_stats['replayed'] >= (5 if os.environ.get('NEMO_GRAPHS', '1') != '0' else 0)


def test_full_batch_properties_at_benchmark_size(tmp_path):
    """Size-independent properties at the BASELINE configuration (8 x 300 full batch, N = 2400):
    (i) the full-batch step equals the same indices passed explicitly, (ii) the per-view
    decomposition of the keypoint loss, (iii) losses are finite and decrease over a few steps,
    (iv) get_preds shapes."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    V, T = 8, 300
    args = syn.published_args(batch_size=512, out_dir=str(tmp_path))
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=syn.make_smpl_assets(6890, seed=1, skin_nnz=4),
               vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    ld_full, info = m.step(None, None, update=False, full_batch=True)
    vi, fi = m.full_indices()
    ld_idx, _ = m.step(vi.cpu(), fi.cpu(), update=False)
    for k in ld_full:
        assert rel_err(ld_idx[k], ld_full[k]) < 1e-5, k      # atomics: summation order may differ
    la, gt = info['loss_all'], info['points2d_gt']
    per_view = torch.stack([(la[vi == v] * gt[vi == v][..., -1:]).mean() for v in range(V)]).mean()
    assert rel_err(ld_full['kp_loss'], per_view) < 1e-5
    first = float(ld_full['total_loss'])
    for _ in range(5):
        ld, _ = m.step(None, None, update=True, full_batch=True)
        assert np.isfinite(ld['total_loss'])
    assert float(ld['total_loss']) < first
    p = m.get_preds()
    assert p['v'].shape == (V, T, 6890, 3) and p['j'].shape == (V, T, 25, 3) and p['poses'].shape == (V, T, 69)
    p2 = m.learned_camera_projection(p['j'].reshape(V * T, 25, 3), vi)
    assert rel_err(p2, info['points2d']) > 0      # parameters moved
    assert torch.isfinite(p2).all()


@pytest.mark.parametrize('version', [0, 1, 2, 3, 4])
def test_script_surface_on_a_constructed_model(version, tmp_path):
    """Every attribute scripts/learned_multi_view_recon_nn.py:192-335 uses resolves on a constructed NemoV0..V4; the
    unconditional rendering call of :199 warns and returns; eval_2d / eval_3d(dynamic_only) write the three CSVs of
    :333-335; other render_* methods say that rendering is out of scope; a run saved by one model is re-opened from
    its model_config.p by the constructor (:155-192)."""
    from test_abi_and_host import SCRIPT_ATTRIBUTES, SCRIPT_METHODS
    from nemo_cvpr2023_amd.neural_motion_model import NEMO_VERSIONS
    V, T = 2, 8
    base = syn.published_args if version >= 2 else syn.default_v1_args
    run = tmp_path / 'run'
    args = base(h_dim=32, monotonic_network_n_nodes=12, batch_size=6, out_dir=str(run), weight_vp_loss=0 if version == 0 else 10)
    if version >= 2:
        args.phase_rbf_dim = 8
    args.model_version = version
    seqs = syn.SyntheticSequences(V, T, seed=3, with_eval=True)
    kw = dict(smpl_assets=syn.make_smpl_assets(128, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    m = NEMO_VERSIONS[version](args, seqs, DEV, **kw)
    assert m.to(DEV) is m
    for name in SCRIPT_METHODS + SCRIPT_ATTRIBUTES:
        assert getattr(m, name) is not None, name
    assert len(m.phase_networks) == m.num_views == V and m.num_frames == T
    assert all('lr' in o.param_groups[0] for o in m.optimizers)
    with pytest.warns(UserWarning, match='rendering'):
        assert m.render_rollout_keypoint_figure(str(tmp_path / 'rollout_keypoint.png'), num_frames=5, num_views=3) is None
    with pytest.raises(NotImplementedError, match='out of scope'):
        m.render_rollout_figure(str(tmp_path / 'x.png'), num_frames=5, num_views=3)
    with pytest.raises(AttributeError):
        m.no_such_attribute
    m.step(*m.draw_batch())
    m.eval_2d(str(run))
    m.eval_3d(str(run))
    m.eval_3d(str(run), dynamic_only=True)
    for fn in ('eval_2d.csv', 'eval_3d.csv', 'eval_3d_dynamic.csv', 'model_config.p'):
        assert (run / fn).exists(), fn
    (run / 'ckpt').mkdir()
    m.save(str(run / 'ckpt' / 'sd_000000.pt'))
    # --test / resume: other hyper-parameters on the command line, the saved ones win
    args2 = base(h_dim=16, monotonic_network_n_nodes=5, batch_size=6, out_dir=str(tmp_path / 'resumed'),
                 load_ckpt_path=str(run / 'ckpt' / 'sd_000000.pt'))
    m2 = NEMO_VERSIONS[version](args2, seqs, DEV, **kw)
    assert m2.args.h_dim == 32 and m2.engine.K == 12
    m2.load(args2.load_ckpt_path)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert (tmp_path / 'resumed' / 'model_config.p').exists()


def test_checkpoint_roundtrip_and_api(tmp_path):
    m, g = build_hip_case('v2_small', tmp_path=tmp_path)
    torch.manual_seed(2)
    vi, fi = m.draw_batch()
    m.step(vi, fi)
    m.step(vi, fi)
    path = str(tmp_path / 'sd.pt')
    m.save(path)
    ld_a, _ = m.step(vi, fi)
    m2, _ = build_hip_case('v2_small', tmp_path=tmp_path)
    m2.load(path)
    ld_b, _ = m2.step(vi, fi)
    for k in ld_a:
        assert rel_err(ld_b[k], ld_a[k]) < 1e-5, k      # resumed run continues identically
    sd = torch.load(path, weights_only=False)
    assert set(sd) == {'model_sd', 'opt_sd'} and len(sd['opt_sd']) == 4
    assert 'learned_motion.net.net.0.weight' in sd['model_sd'] and 'phase_rbf.centres' in sd['model_sd']
    st = sd['opt_sd'][1]['state']
    assert float(st[0]['step']) == 2.0 and st[0]['exp_avg'].shape == (48, 21)
    # phase network modules are callable (script plots them, :317-328)
    y = m.phase_networks[0](torch.linspace(0, 1, 50).unsqueeze(1))
    assert y.shape == (50, 1) and float(y[0]) == 0.0 and abs(float(y[-1]) - 1.0) < 1e-4
    assert (y[1:] >= y[:-1] - 1e-6).all()
    lrs = [o.param_groups[0]['lr'] for o in m.optimizers]
    assert lrs == [0.1, 1e-4, 1e-4, 1e-3]


# ------------------------------------------------------------------------------------------ next rows (8f)
def _hip_from_init(g, args, seqs, num_verts=128):
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    m = NemoV2(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(num_verts, seed=1),
               vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    state = {k[len('init__'):].replace('__', '.'): torch.tensor(v) for k, v in g.items() if k.startswith('init__')}
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not missing and not unexpected
    return m


def test_eval_metrics_on_device_match_reference_csvs(tmp_path):
    """nemo_cvpr2023_amd/evaluation.py (HIP kernels) == eval_2d.csv / eval_3d.csv / eval_3d_dynamic.csv written
    by the real reference (tests/golden/eval_eval_v2.npz)."""
    from nemo_cvpr2023_amd import evaluation as ev
    g = load_golden('eval_eval_v2')
    V, T = int(g['meta__V']), int(g['meta__T'])
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=8, out_dir='', phase_rbf_dim=16)
    args.model_version = 2
    m = _hip_from_init(g, args, syn.SyntheticSequences(V, T, seed=1234, with_eval=True))
    res = ev.evaluate_all(m, str(tmp_path))
    for name in ('eval_2d', 'eval_3d', 'eval_3d_dynamic'):
        assert res[name], name
        for k, v in res[name].items():
            assert rel_err(np.asarray(v), g[f'{name}__{k}']) < 1e-4, (name, k)
        rows = list(__import__('csv').reader(open(tmp_path / (name + '.csv'))))
        assert rows[0][1:] == list(res[name]) and len(rows) == V + 1


def test_fit_driver_on_device_reproduces_script_order(tmp_path):
    """fit.run_fit over the HIP model == the real reference driven in the script's order; checkpoints land
    where the script puts them."""
    from nemo_cvpr2023_amd.fit import run_fit
    g = load_golden('script_script_v2')
    V, T, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16,
                              n_steps=len(g['total_loss']), warmup_step=int(g['meta__n_warm']),
                              opt_cam_step=int(g['meta__n_cam']))
    args.model_version = 2
    m = _hip_from_init(g, args, syn.SyntheticSequences(V, T, seed=1234))
    torch.manual_seed(2)
    res = run_fit(m, args, out_dir=str(tmp_path))
    assert rel_err(np.float32(res['init']['total_loss']), g['init_total_loss']) < 1e-4
    assert rel_err(np.asarray(res['warmup_losses']), g['warmup_losses']) < 1e-4
    assert rel_err(np.asarray(res['cam_losses']), g['cam_losses']) < 1e-4
    assert rel_err(np.asarray(res['losses']['total_loss']), g['total_loss']) < 1e-4
    assert rel_err(np.asarray(res['losses']['kp_loss']), g['kp_loss_pure']) < 1e-4
    assert rel_err(np.asarray([float(x['total_loss']) for x in res['evals'].values()]), g['eval_total_loss']) < 1e-4
    assert (tmp_path / 'ckpt' / 'sd_000000.pt').exists() and (tmp_path / 'info' / '_init.pt').exists()


def test_loss_curve_parity_200_steps():
    """loss-curve parity over 200 optimisation steps from the same seed, published-run structure
    (NemoV2, all loss terms, minibatches drawn in the script's order) on a small mesh: the HIP model and the
    CPU oracle start from the same state and see the same batches; their total-loss curves must stay
    together (free-running, no re-synchronisation -- Adam's amplification of rounding noise included)."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T, B, steps = 4, 40, 64, 200
    args = syn.published_args(h_dim=64, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(2)
    cm, co = [], []
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)          # ~2000 tiny CPU ops per oracle step: a big thread pool only adds hand-off cost
    try:
        for _ in range(steps):
            vi = torch.randint(0, V, (B,), generator=gen)
            fi = torch.randint(0, T, (B,), generator=gen)
            cm.append(float(m.step(vi, fi)[0]['total_loss']))
            co.append(float(o.step(vi, fi)[0]['total_loss']))
    finally:
        torch.set_num_threads(nthreads)
    cm, co = np.asarray(cm), np.asarray(co)
    rel = np.abs(cm - co) / np.abs(co)
    assert rel[:5].max() < 1e-4, rel[:5]                 # identical inputs, identical state: the 1e-4 gate
    assert rel.max() < 2e-3 and rel.mean() < 5e-4, (rel.max(), rel.mean())
    assert co[-20:].mean() < co[:20].mean()              # and the fit actually descends


def test_loss_curve_parity_real_size():
    """the free-running comparison at the REAL sizes of the published run (h = 1000, RBF 100, 6890-vertex
    mesh, every loss term): minibatches of 512 drawn from 8 x 300 in the script's order, HIP model and CPU oracle from
    the same state, no re-synchronisation.  40 steps in the default suite; NEMO_LONG_TESTS=1 runs the full 200 steps
    (about three minutes of oracle time; the curve of that run is committed as profiles/r02_loss_curve_200.json).
    Measured (profiles/r02_loss_curve_200.json, maxima per 20-step window): every loss term agrees to 1e-6 over the
    first 20 steps, to 1e-4 through step 80 (vp_recon 1.8e-4), then the two free-running fits separate slowly -- Adam
    turns rounding-level differences of noise-level gradients into +-lr moves -- to 2e-3 (kp, total) / 1e-2
    (vp_recon) by step 200.  Gate: 1e-4 on every term over the first 40 steps (3e-4 up to step 80), and the envelope
    max 3e-2 / mean 5e-3 on every term over the whole run; the fit descends."""
    import json
    import os
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T, B = 8, 300, 512
    steps = 200 if os.environ.get('NEMO_LONG_TESTS') else 40
    args = syn.published_args(batch_size=B, out_dir='')
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(2)
    keys = ('kp_loss', 'gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss')
    cm, co = {k: [] for k in keys}, {k: [] for k in keys}
    for _ in range(steps):
        vi = torch.randint(0, V, (B,), generator=gen)
        fi = torch.randint(0, T, (B,), generator=gen)
        lm, lo = m.step(vi, fi)[0], o.step(vi, fi)[0]
        for k in keys:
            cm[k].append(float(lm[k]))
            co[k].append(float(lo[k]))
    rel = {k: np.abs(np.asarray(cm[k]) - np.asarray(co[k])) / np.maximum(np.abs(np.asarray(co[k])), 1e-6) for k in keys}
    out = os.environ.get('NEMO_CURVE_OUT')
    if out:
        json.dump({'steps': steps, 'batch': B, 'hip': cm, 'oracle': co,
                   'rel_max': {k: float(v.max()) for k, v in rel.items()},
                   'rel_mean': {k: float(v.mean()) for k, v in rel.items()}}, open(out, 'w'))
    for k in keys:
        assert rel[k][:40].max() < 1e-4, (k, float(rel[k][:40].max()))
        assert rel[k][:80].max() < 3e-4, (k, float(rel[k][:80].max()))
        assert rel[k].max() < 3e-2 and rel[k].mean() < 5e-3, (k, float(rel[k].max()), float(rel[k].mean()))
    assert np.mean(co['total_loss'][-10:]) < np.mean(co['total_loss'][:10])


def test_200_steps_of_the_published_configuration_in_lock_step():
    """SURVEY 8(d)'s "200 steps", GATED: the published configuration at its real sizes (NemoV2,
    h = 1000, RBF 100, 6890 vertices, every loss term, minibatches of 512 out of 8 x 300 in the script's order) runs 200
    update steps on the HIP path; before every 10th step -- and before each of the first five -- the CPU oracle takes
    over the HIP model's parameters and Adam state and both take THAT step: every loss term of every compared step must
    agree to 1e-4 (25 oracle steps instead of 200: the suite's CPU budget), i.e. the step function is held to the
    reference along the whole trajectory the HIP fit actually takes, and the fit descends."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T, B, steps = 8, 300, 512, 200
    args = syn.published_args(batch_size=B, out_dir='')
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.cpu() for k, v in m.state_dict().items()})
    gen = torch.Generator().manual_seed(2)
    keys = ('kp_loss', 'gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss')
    worst, compared, curve = 0.0, 0, []
    for s in range(steps):
        vi = torch.randint(0, V, (B,), generator=gen)
        fi = torch.randint(0, T, (B,), generator=gen)
        check = s < 5 or s % 10 == 9
        if check:
            o.load_state({k: v.cpu() for k, v in m.state_dict().items()})
            for oo, mo in zip(o.optimizers, m.optimizers):
                sd = mo.state_dict()
                if sd['state']:
                    oo.load_state_dict(sd)
        ld_h = m.step(vi, fi)[0]
        curve.append(float(ld_h['total_loss']))
        if check:
            ld_o = o.step(vi, fi)[0]
            compared += 1
            for k in keys:
                err = abs(float(ld_h[k]) - float(ld_o[k])) / max(abs(float(ld_o[k])), 1e-6)
                worst = max(worst, err)
                assert err <= 1e-4, (s, k, float(ld_h[k]), float(ld_o[k]))
    assert compared == 25
    assert np.mean(curve[-10:]) < np.mean(curve[:10])


def test_bf16_loss_curve_stays_with_the_fp32_curve_over_100_steps():
    """the bf16 build (operands bf16 in memory, fp32 accumulation, fp32 master weights) against the
    fp32 build over 100 FREE-RUNNING update steps at the published sizes (minibatches of 512 out of 8 x 300, same draws,
    same initial state): the bf16 fit must descend like the fp32 one -- every loss term within 2 % of the fp32 curve at
    every step, the mean total loss of the last ten steps within 1 %."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    V, T, B, steps = 8, 300, 512, 100
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
    models = []
    for dt in ('f32', 'bf16'):
        args = syn.published_args(batch_size=B, out_dir='')
        args.gemm_dtype = dt
        torch.manual_seed(0)
        models.append(NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm))
    models[1].load_state_dict({k: v.clone() for k, v in models[0].state_dict().items()}, strict=False)
    assert models[1].engine.b16mem and not models[0].engine.bf16
    gen = torch.Generator().manual_seed(2)
    keys = ('kp_loss', 'gmm_loss', 'vp_recon_loss', 'vp_kl_loss', 'total_loss')
    cur = [{k: [] for k in keys} for _ in models]
    for _ in range(steps):
        vi = torch.randint(0, V, (B,), generator=gen)
        fi = torch.randint(0, T, (B,), generator=gen)
        for m_, c_ in zip(models, cur):
            ld = m_.step(vi, fi)[0]
            for k in keys:
                c_[k].append(float(ld[k]))
    for k in keys:
        a, b = np.asarray(cur[0][k]), np.asarray(cur[1][k])
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
        assert rel.max() < 2e-2, (k, float(rel.max()), int(rel.argmax()))
    f32_end, b16_end = np.mean(cur[0]['total_loss'][-10:]), np.mean(cur[1]['total_loss'][-10:])
    assert abs(b16_end - f32_end) <= 1e-2 * abs(f32_end), (f32_end, b16_end)
    assert b16_end < np.mean(cur[1]['total_loss'][:10])


def test_optional_temporal_smoothness_term():
    """SURVEY 8f-4 (an extension, off by default): with args.weight_smooth > 0 a full-batch step adds
    w * 0.5 * sum |J[v,t+1] - J[v,t]|^2 over the 25 output joints; value and every parameter gradient must
    match the oracle's autograd, and minibatch steps / weight 0 must be untouched."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T = 3, 9
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=8, out_dir='', phase_rbf_dim=16)
    args.weight_smooth = 50.0
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    with torch.no_grad():                                   # make the motion non-trivial
        for k, p in m.named_parameters():
            if 'learned_motion' in k and 'weight' in k:
                p.add_(0.05 * torch.randn_like(p))
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.cpu() for k, v in m.state_dict().items()})
    lm, _ = m.step(None, None, update=True, full_batch=True)
    lo, _ = o.step(None, None, update=True, full_batch=True)
    assert float(lo['smooth_loss']) > 0
    for k in lo:
        assert rel_err(lm[k], lo[k]) < 1e-4, (k, lm[k], lo[k])
    named = dict(m.named_parameters())
    for k, p in o.P.items():
        if k != 'learned_betas' and p.grad is not None and float(p.grad.abs().max()) > 0:
            assert rel_err(named[k].grad, p.grad) < 2e-3, k
    vi, fi = torch.randint(0, V, (8,)), torch.randint(0, T, (8,))
    assert 'smooth_loss' not in m.step(vi, fi)[0] and 'smooth_loss' not in o.step(vi, fi)[0]


@pytest.mark.parametrize('version', [2, 3])
def test_early_loss_readback_equals_synchronous_readback(version):
    """The step hands its loss scalars to the host through pinned memory as soon as they are final
    (engine.publish_scalars) while backward + Adam still run.  Losses, per-step info and the state
    read right after every step must equal those of the synchronous read-back (up to the run-to-run
    rounding of the atomically reduced loss terms), in eager and in graph mode, for minibatch and
    full-batch steps."""
    from nemo_cvpr2023_amd.neural_motion_model import NEMO_VERSIONS
    V, T, B = 3, 12, 16
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16)
    if version == 3:
        args.weight_instance_loss, args.weight_3d_loss, args.instance_code_size = 0.5, 1.0, 5
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm()
    runs = []
    for early in (True, False):
        torch.manual_seed(0)
        m = NEMO_VERSIONS[version](args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
        m.engine.early_readback = early
        gen = torch.Generator().manual_seed(5)
        log = []
        for it in range(8):
            if it % 3 == 2:
                ld, info = m.step(None, None, update=True, full_batch=True)
            else:
                vi, fi = torch.randint(0, V, (B,), generator=gen), torch.randint(0, T, (B,), generator=gen)
                ld, info = m.step(vi, fi)
            sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}    # right after the step
            log.append(({k: np.asarray(v).copy() for k, v in ld.items()}, info['loss_all'].cpu().numpy(), sd))
        ld, info = m.step(None, None, update=False, full_batch=True)
        log.append(({k: np.asarray(v).copy() for k, v in ld.items()}, info['loss_all'].cpu().numpy(), {}))
        runs.append(log)
    for (la, ia, sa), (lb, ib, sb) in zip(*runs):
        assert la.keys() == lb.keys()
        for k in la:
            assert rel_err(la[k], lb[k]) < 1e-4, (k, la[k], lb[k])
        assert rel_err(ia, ib) < 1e-4
        for k in sa:         # (Adam turns rounding noise on noise-level gradients into +-lr steps: sanity bound)
            assert rel_err(sa[k], sb[k]) < 2e-2, k
    for log in runs:         # a state read right after step k already holds update k
        for (_, _, s0), (_, _, s1) in zip(log[:-2], log[1:-1]):
            k = 'learned_motion.net.net.2.weight'
            assert not torch.equal(s0[k], s1[k])


def test_failed_graph_capture_falls_back_to_uncaptured_launches(monkeypatch):
    """If the HIP-graph capture of a step variant fails (e.g. invalidated by another thread), the variant keeps
    running kernel by kernel through the same engine and the results do not change."""
    if os.environ.get('NEMO_GRAPHS', '1') == '0':
        pytest.skip('NEMO_GRAPHS=0: nothing is captured')
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    V, T, B = 2, 8, 8
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=B, out_dir='', phase_rbf_dim=16)
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(128, seed=1), syn.make_vposer_state(), syn.make_gmm()

    class Boom:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            raise RuntimeError('capture invalidated (test)')

        def __exit__(self, *a):
            return False
    runs = []
    for broken in (False, True):
        torch.manual_seed(0)
        m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
        if broken:
            monkeypatch.setattr(torch.cuda, 'graph', Boom)
        with pytest.warns(UserWarning) if broken else __import__('contextlib').nullcontext():
            runs.append([{k: float(v) for k, v in m.step(None, None, update=True, full_batch=True)[0].items()}
                         for _ in range(5)])
        if broken:
            assert 'eager' in m.engine._ws(V * T)['graphs'].values()
    for a, b in zip(*runs):
        for k in a:
            assert rel_err(a[k], b[k]) < 1e-4, (k, a[k], b[k])


def test_checkpoint_written_by_the_reference_loads_into_the_hip_model():
    """`load()` (:268-280) of a file produced by the reference's own `save()` (:257-266) -- model keys, four
    torch-format Adam states with their step counts -- and the run the reference itself continued from it."""
    import os
    from conftest import GOLDEN
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from test_oracle_golden import _ckpt_case, check_resumed_run
    g, V, T, B, args, seqs = _ckpt_case()
    torch.manual_seed(321)                                     # a different init: everything must come from the file
    m = NemoV2(args, seqs, DEV, smpl_assets=syn.make_smpl_assets(128, seed=1), vposer_state=syn.make_vposer_state(),
               gmm=syn.make_gmm())
    m.load(os.path.join(GOLDEN, 'ckpt_ref_v2.pt'))
    assert [float(next(iter(o.state_dict()['state'].values()))['step']) for o in m.optimizers] == [3.0, 5.0, 5.0, 3.0]
    check_resumed_run(m.step, g)
    sd = m.state_dict()
    for k in ('learned_cameras', 'learned_motion.net.net.2.weight', 'learned_motion.rot_out.bias',
              'phase_networks.1.scales', 'learned_instance_code'):
        assert rel_err(sd[k], g['final__' + k.replace('.', '__')]) < 5e-3, k


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_two_fits_from_one_seed_are_bit_identical(dtype):
    """every cross-block sum of the fit (per-view loss / camera-gradient sums of the key-point kernels,
    phase-network gradients, bias column sums, loss scalars) is added in a fixed order by a launch's last-arriving block
    instead of by float atomics (csrc/common.h, include/nemo_hip.h nemo_reduce_scratch_reset).  The published schedule's three
    phases -- 30 warm-up, 100 camera-fit and 100 minibatch-512 iterations at the published sizes (NemoV2, 8 x 300, h = 1000,
    6890 vertices, every loss term) -- run twice from one seed on freshly built models: every loss of every iteration and every
    parameter and Adam moment at the end must be BIT-identical (rounds 1 - 4: final camera loss 3139 vs 3273 between runs)."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from nemo_cvpr2023_amd.fit import run_fit
    V, T = 8, 300
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
    runs = []
    for rep in range(2):
        args = syn.published_args(batch_size=512, out_dir='', n_steps=100, warmup_step=30, opt_cam_step=100)
        args.gemm_dtype = dtype
        torch.manual_seed(0)
        m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
        torch.manual_seed(0)
        res = run_fit(m, args, out_dir=None, evaluate=None)
        torch.cuda.synchronize()
        e = m.engine
        runs.append((res, {k: v.detach().clone() for k, v in m.state_dict().items()}, e.exp_avg.clone(), e.exp_avg_sq.clone()))
    (ra, sa, ma, va), (rb, sb, mb, vb) = runs
    assert ra['warmup_losses'] == rb['warmup_losses'] and len(ra['warmup_losses']) == 30
    assert ra['cam_losses'] == rb['cam_losses'] and len(ra['cam_losses']) == 100
    for k in ra['losses']:
        assert ra['losses'][k] == rb['losses'][k], k
    assert len(ra['losses']['total_loss']) == 100
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(ma, mb) and torch.equal(va, vb)
    assert np.isfinite(ra['losses']['total_loss'][-1]) and ra['cam_losses'][-1] < ra['cam_losses'][0]


def test_two_engines_on_two_streams_and_threads_step_concurrently_and_reproducibly():
    """ABI 17: the arena of the ordered reductions is owned by each engine and bound to the calling host thread
    (include/nemo_hip.h nemo_reduce_ws_bind) -- no process-global scratch.  Two models step CONCURRENTLY from two host threads on two
    streams (minibatch and full-batch steps, replayed graphs); each must reproduce, bit for bit, what a third model does alone, and
    no launch may have fallen back to float atomics (nemo_reduce_fallbacks)."""
    import threading
    from nemo_cvpr2023_amd import _lib
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    V, T, steps = 4, 60, 24
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(512, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()

    def build():
        args = syn.published_args(batch_size=64, out_dir='', h_dim=256)
        torch.manual_seed(0)
        return NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, V, (64,), generator=gen), torch.randint(0, T, (64,), generator=gen)) for _ in range(steps)]

    def run(m, out, stream=None):
        with torch.cuda.stream(stream or torch.cuda.current_stream()):
            for i, (vi, fi) in enumerate(batches):
                ld, _ = m.step(None, None, update=True, full_batch=True) if i % 3 == 2 else m.step(vi, fi, update=True)
                out.append({k: float(v) for k, v in ld.items()})
            torch.cuda.current_stream().synchronize()
    L = _lib.load()
    fb0 = int(L.nemo_reduce_fallbacks())
    solo, ref = build(), []
    run(solo, ref)
    ma, mb, oa, ob = build(), build(), [], []
    assert ma.engine.red_ws.data_ptr() != mb.engine.red_ws.data_ptr()
    # the first steps one after the other (graph captures are not meant to overlap), the rest side by side
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    errs = []

    def guarded(*a):
        try:
            run(*a)
        except Exception as ex:        # noqa: BLE001
            errs.append(ex)
    head, tail = batches[:6], batches[6:]
    batches[:] = head
    run(ma, oa, sa)
    run(mb, ob, sb)
    batches[:] = tail
    ta, tb = threading.Thread(target=guarded, args=(ma, oa, sa)), threading.Thread(target=guarded, args=(mb, ob, sb))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    torch.cuda.synchronize()
    assert oa == ref and ob == ref
    for k, v in solo.state_dict().items():
        assert torch.equal(v, ma.state_dict()[k]) and torch.equal(v, mb.state_dict()[k]), k
    assert int(L.nemo_reduce_fallbacks()) == fb0


@pytest.mark.parametrize('B', [512, 2400])
def test_every_gradient_at_the_true_init_with_the_mesh_term_on(B):
    """the configuration bench.py actually times -- sparse skinning (skin_nnz = 4), mesh_blend 'f32_split', the
    split-precision MotionNet chain (2400 rows), the PUBLISHED initialisation (no x 2e3 on rot_out: pose features ~1e-3 ... 1e-5, the
    low fp16 pieces subnormal) and the v2v term ON -- held on EVERY parameter gradient of one update step: within 1e-4 of the tensor's
    largest entry + the oracle's own fp32-vs-float64 distance + what sign(0) ties of the L1 term can change (tiebound.py)."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    from tiebound import model_v2v_tie_bound
    V, T = 8, 300
    args = syn.published_args(batch_size=512, out_dir='')
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    e = m.engine
    assert e.mesh_split and e.ctx.skin_sparse and e.ctx.split_ok and e.mlp_split
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    torch.manual_seed(4)
    if B == 2400:
        vi, fi = o.full_indices()
        call = lambda mdl: mdl.step(None, None, update=True, full_batch=True)
    else:
        vi, fi = torch.randint(0, V, (B,)), torch.randint(0, T, (B,))
        call = lambda mdl: mdl.step(vi, fi, update=True)
    o64 = _float64_twin(o)
    bound, n_ties = model_v2v_tie_bound(o64, vi, fi)
    torch.set_default_dtype(torch.float64)
    try:
        call(o64)
    finally:
        torch.set_default_dtype(torch.float32)
    ld_o, _ = call(o)
    ld_h, _ = call(m)
    for k in ld_o:
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4 or abs(float(ld_o[k])) < 1e-12, (k, ld_h[k], ld_o[k])
    keys = [k for k, p in o.P.items() if k != 'learned_betas' and p.grad is not None]
    assert len(keys) >= 28
    zero = torch.zeros(())
    full_bound = {k: bound.get(k, zero) for k in keys}
    worst = _assert_gradients_up_to_l1_ties(m, o, o64, full_bound, keys)
    print(f'B = {B}: {n_ties} L1 coordinates within rounding of a tie; worst gradient error beyond the allowances / scale: {worst:.3g}')


def test_locality_body_model_step_vs_oracle():
    """a step on the spatially structured synthetic body model (synthetic.make_smpl_assets(locality=True):
    vertices ordered by body part, 1 - 2 dominant skinning weights, sparse local joint regressors -- the access patterns of the
    published SMPL model) against the oracle: losses, joints and the gradients of the shared tensors."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    from tiebound import model_v2v_tie_bound
    V, T, B = 8, 300, 512
    args = syn.published_args(batch_size=B, out_dir='')
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4, locality=True), syn.make_vposer_state(), syn.make_gmm()
    assert int((assets['lbs_weights'] > 0).sum(1).max()) <= 4 and int((assets['J_regressor'] > 0).sum(1).max()) < 200
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    assert m.engine.ctx.skin_sparse and m.engine.ctx.split_ok
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    o = OracleNemo(2, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m.state_dict().items()})
    torch.manual_seed(2)
    vi, fi = torch.randint(0, V, (B,)), torch.randint(0, T, (B,))
    o64 = _float64_twin(o)
    bound, n_ties = model_v2v_tie_bound(o64, vi, fi)
    torch.set_default_dtype(torch.float64)
    try:
        o64.step(vi, fi, update=True)
    finally:
        torch.set_default_dtype(torch.float32)
    ld_o, info_o = o.step(vi, fi, update=True)
    ld_h, info_h = m.step(vi, fi, update=True)
    for k in ld_o:
        assert rel_err(ld_h[k], ld_o[k]) < 1e-4, (k, ld_h[k], ld_o[k])
    assert rel_err(info_h['j'], info_o['j']) < 1e-4 and rel_err(info_h['points2d'], info_o['points2d']) < 1e-4
    _assert_gradients_up_to_l1_ties(m, o, o64, bound, ('learned_motion.net.net.2.weight', 'learned_motion.rot_out.weight',
                                                       'learned_cameras', 'phase_rbf.log_sigmas', 'learned_instance_code'))
