"""CPU-side checks: the C-ABI library loads and exports every declared symbol (no compute calls),
the flat parameter layout mirrors the reference optimiser order, the product never imports the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'nemo_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nemo_[A-Za-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from nemo_cvpr2023_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/nemo_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    header = open(os.path.join(ROOT, 'include', 'nemo_hip.h')).read()
    assert lib.nemo_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define NEMO_ABI_VERSION (\d+)', header).group(1))


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: bad arguments return <0 on any box."""
    from nemo_cvpr2023_amd import _lib
    lib = _lib.load()
    assert lib.nemo_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, 0, 1.0, 0, 1, None, 0, None) < 0
    assert lib.nemo_rot6d_fwd(4, 0, None, 0, 1, None, None, None) < 0
    assert lib.nemo_adam_step(99, None, None, None, None, None, 0.9, 0.999, 1e-8, None) < 0
    assert lib.nemo_ctx_num_verts(None) == -1


def test_param_layout_matches_reference_optimizer_order():
    from nemo_cvpr2023_amd.engine import ParamLayout
    lay = ParamLayout(V=3, K=20, D=16, C=5, h=48, din=21)
    # 'comm' = 8 floats that ride the shared-gradient all-reduce; no optimiser owns them
    assert list(lay.groups) == ['cameras', 'motion', 'comm', 'phase', 'instance']
    assert lay.groups['motion'] == [
        'learned_motion.net.net.0.weight', 'learned_motion.net.net.0.bias',
        'learned_motion.net.net.2.weight', 'learned_motion.net.net.2.bias',
        'learned_motion.net.net.4.weight', 'learned_motion.net.net.4.bias',
        'learned_motion.rot_out.weight', 'learned_motion.rot_out.bias',
        'learned_motion.linear_out.weight', 'learned_motion.linear_out.bias', 'phase_rbf.log_sigmas']
    assert lay.groups['phase'][:4] == ['phase_networks.0.shifts', 'phase_networks.0.scales',
                                       'phase_networks.1.shifts', 'phase_networks.1.scales']
    # non-overlapping in MEMORY order (entries), every tensor on a 16-byte boundary with < 4 floats of
    # padding in front; the two MLP heads sit next to each other
    end = 0
    for name, (off, shape) in lay.entries.items():
        assert off % 4 == 0 and 0 <= off - end < 4, name
        end = off + int(np.prod(shape))
    assert 0 <= lay.total - end < 4
    e = lay.entries
    assert e['learned_motion.linear_out.weight'][0] == e['learned_motion.rot_out.weight'][0] + 144 * 48
    assert e['learned_motion.linear_out.bias'][0] == e['learned_motion.rot_out.bias'][0] + 144
    assert e['learned_motion.rot_out.bias'][0] == e['learned_motion.linear_out.weight'][0] + 3 * 48
    a, b = lay.span(lay.groups['motion'])
    assert b - a == 48 * 21 + 48 + 2 * (48 * 48 + 48) + 144 * 48 + 144 + 3 * 48 + 3 + 16 + 8   # (8 comm scalars inside)
    a2, b2 = lay.span(lay.groups['motion'] + lay.groups['comm'])
    assert (a2, b2) == (a, b)              # one contiguous all-reduce slice: the 8 loss scalars sit inside it
    # ... cut into three contiguous gradient buckets in the order the backward completes them (dist.py):
    # [heads + layer 4] is the LAST slice in memory, [layer 0 + RBF widths + loss scalars] the first
    bk = lay.buckets()
    assert bk[2][0] == a and bk[0][1] == b and bk[2][1] <= bk[1][0] <= bk[1][1] <= bk[0][0]
    assert e['_comm_scalars'][0] >= bk[2][0] and e['_comm_scalars'][0] + 8 <= bk[2][1]
    assert e['learned_motion.rot_out.bias'][0] >= bk[0][0] and e['learned_motion.net.net.2.weight'][0] == bk[1][0]
    # the 16-byte alignment must hold for a one-view shard too (9 camera floats in front of the MLP)
    lay1 = ParamLayout(V=1, K=20, D=16, C=5, h=48, din=21)
    assert all(off % 4 == 0 for off, _ in lay1.entries.values())


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nemo_cvpr2023_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), fn


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nemo_cvpr2023_amd import synthetic as syn
    from nemo_cvpr2023_amd._lib import NemoHipError
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    args = syn.published_args(h_dim=16, out_dir='')
    with pytest.raises(NemoHipError):
        NemoV2(args, syn.SyntheticSequences(2, 4), 'cuda:0', smpl_assets=syn.make_smpl_assets(64),
               vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())


def test_vposer_folding_matches_unfolded():
    """BatchNorm folding done on the host for the HIP path equals the oracle's eval-mode encoder."""
    import torch
    from nemo_cvpr2023_amd import synthetic as syn
    from nemo_cvpr2023_amd.engine import fold_vposer
    from oracle import ops
    sd = syn.make_vposer_state()
    f = fold_vposer(sd, 'cpu')
    x = 0.3 * torch.randn(7, 63, generator=torch.Generator().manual_seed(0))
    lin = torch.nn.functional.linear
    h = torch.nn.functional.leaky_relu(lin(x, f['e2w'], f['e2b']))
    ml = lin(h, f['emw'], f['emb'])          # BN + Linear6 + Linear7 + heads folded into one 512 -> 64 map
    mean, scale = ops.VPoserOracle(sd).encode(x)
    assert float((ml[:, :32] - mean).abs().max()) < 1e-5
    assert float((torch.nn.functional.softplus(ml[:, 32:]) - scale).abs().max()) < 1e-5
    # decode(q_z.mean): the decoder's first Linear composed with the mean head (d0mw, d0mb) == the two maps one after the
    # other, and the whole decoder from there equals the oracle's decode(mean)
    d1 = torch.nn.functional.leaky_relu(lin(ml[:, :32], f['d0w'], f['d0b']))
    d1c = torch.nn.functional.leaky_relu(lin(h, f['d0mw'], f['d0mb']))
    assert float((d1c - d1).abs().max()) < 1e-5 * float(d1.abs().max())
    d3 = lin(torch.nn.functional.leaky_relu(lin(d1c, f['d3w'], f['d3b'])), f['d5w'], f['d5b'])
    _, Rdec = ops.VPoserOracle(sd).decode(mean)
    ref6 = Rdec.reshape(7, 21, 3, 3)
    got = ops.rot6d_to_rotmat(d3.reshape(-1, 6)).reshape(7, 21, 3, 3)
    assert float((got - ref6).abs().max()) < 1e-5


def test_real_asset_loaders_round_trip(tmp_path):
    """assets.py reads the file formats the reference reads from software/ (SMPL .pkl with a sparse joint
    regressor and (NV,3,207) pose blend shapes, a VPoser snapshot with the 'vp_model.' prefix, gmm_08.pkl);
    files are written here from the synthetic assets and must come back identical."""
    import pickle
    import scipy.sparse as sp
    import torch
    from nemo_cvpr2023_amd import assets, synthetic as syn
    a = syn.make_smpl_assets(64, seed=3)
    nv = a['v_template'].shape[0]
    smpl_dir = tmp_path / 'smpl'
    smpl_dir.mkdir()
    d = {'v_template': a['v_template'].numpy(), 'shapedirs': np.concatenate(
            [a['shapedirs'].numpy(), np.zeros((nv, 3, 290), np.float32)], 2),          # 300 shape components on disk
         'posedirs': a['posedirs'].numpy().T.reshape(nv, 3, 207),
         'J_regressor': sp.csc_matrix(a['J_regressor'].numpy()), 'weights': a['lbs_weights'].numpy()}
    with open(smpl_dir / 'basicModel_neutral_lbs_10_207_0_v1.0.0.pkl', 'wb') as f:
        pickle.dump(d, f)
    np.save(tmp_path / 'J_regressor_extra.npy', a['J_regressor_extra'].numpy())
    got = assets.load_smpl_assets(str(smpl_dir), str(tmp_path / 'J_regressor_extra.npy'))
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'J_regressor_extra', 'lbs_weights'):
        assert torch.equal(got[k], a[k].float()), k
    assert got['parents'].tolist() == syn.SMPL_PARENTS and got['joint_map'].tolist() == syn.JOINT_MAP_49
    vp = syn.make_vposer_state()
    (tmp_path / 'V02_05' / 'snapshots').mkdir(parents=True)
    torch.save({'state_dict': {'vp_model.' + k: v for k, v in vp.items()}},
               tmp_path / 'V02_05' / 'snapshots' / 'V02_05_epoch=13.ckpt')
    got = assets.load_vposer_state(str(tmp_path / 'V02_05'))
    assert set(got) == set(vp) and all(torch.equal(got[k], vp[k]) for k in vp)
    g = syn.make_gmm()
    with open(tmp_path / 'gmm_08.pkl', 'wb') as f:
        pickle.dump(g, f)
    got = assets.load_gmm(str(tmp_path))
    assert all(np.array_equal(got[k], g[k]) for k in ('means', 'covars', 'weights'))
    with pytest.raises(FileNotFoundError):
        assets.load_smpl_assets(str(tmp_path / 'nowhere'))


# What scripts/learned_multi_view_recon_nn.py touches on the model object (:192-335) and nemo/utils/render_utils.py:90-158
SCRIPT_METHODS = ['to', 'render_rollout_keypoint_figure', 'step', 'warmup', 'opt_cam', 'save', 'load', 'eval_2d', 'eval_3d',
                  'get_preds', 'get_preds_batch', 'learned_camera_projection', 'state_dict', 'load_state_dict']
SCRIPT_ATTRIBUTES = ['optimizers', 'phase_networks', 'num_views', 'num_frames', 'args', 'device']


def test_script_surface_resolves_on_every_model_class():
    """The one-line swap of INTEGRATION.md section A: every method the reference script calls on the model object exists
    on NemoV0..V4 with the reference's signature (argument names and defaults of eval_2d / eval_3d, :522 / :1056-1061);
    the instance attributes are assigned in the constructor (checked on a constructed model by the GPU suite,
    tests/test_gpu_model.py::test_script_surface_on_a_constructed_model)."""
    import inspect
    from nemo_cvpr2023_amd import neural_motion_model as nm
    for k in range(5):
        cls = nm.NEMO_VERSIONS[k]
        assert cls.__name__ == f'NemoV{k}'
        for name in SCRIPT_METHODS:
            assert callable(getattr(cls, name)), (cls, name)
        sig = inspect.signature(cls.eval_3d)
        assert list(sig.parameters) == ['self', 'out_dir', 'num_frames', 'num_views', 'view_idxs', 'dynamic_only']
        assert sig.parameters['dynamic_only'].default is False and sig.parameters['num_frames'].default == -1
        assert list(inspect.signature(cls.eval_2d).parameters) == ['self', 'out_dir', 'num_frames', 'num_views', 'view_idxs']
        src = inspect.getsource(nm.MultiViewModel)
        for name in SCRIPT_ATTRIBUTES:
            assert re.search(rf'self\.{name}\b[^=\n]*=[^=]', src), name


def test_constructor_takes_the_saved_config_of_a_checkpointed_run(tmp_path):
    """nemo/neural_motion_model.py:155-192: args.load_ckpt_path = <run>/ckpt/sd.pt -> the model is built from the args
    saved in <run>/model_config.p (--test / resumed runs, scripts:313-314); without that file, from the args given."""
    import joblib
    from types import SimpleNamespace
    from nemo_cvpr2023_amd.neural_motion_model import MultiViewModel
    run = tmp_path / 'run'
    (run / 'ckpt').mkdir(parents=True)
    saved = SimpleNamespace(h_dim=123, data_loader_type='generic', nemo_cfg=None, start_phase=0, n_frames=10, run_hmr=False)
    joblib.dump({'args': saved}, str(run / 'model_config.p'))
    seqs = object()
    given = SimpleNamespace(h_dim=7, load_ckpt_path=str(run / 'ckpt' / 'sd_000499.pt'), out_dir=str(tmp_path / 'new'))
    a, s = MultiViewModel._saved_config(given, seqs)
    assert a.h_dim == 123 and a.include_vs is True and a.include_pare is True and s is seqs
    saved.data_loader_type = 'penn_action'
    joblib.dump({'args': saved}, str(run / 'model_config.p'))
    a, _ = MultiViewModel._saved_config(given, seqs)
    assert a.include_vs is True and a.include_pare is False
    saved.data_loader_type = 'nonsense'
    joblib.dump({'args': saved}, str(run / 'model_config.p'))
    with pytest.raises(ValueError):
        MultiViewModel._saved_config(given, seqs)
    (run / 'model_config.p').unlink()
    a, _ = MultiViewModel._saved_config(given, seqs)
    assert a is given
    given.load_ckpt_path = ''
    assert MultiViewModel._saved_config(given, seqs)[0] is given
