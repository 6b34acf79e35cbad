"""Debug aid: phase-network gradient of a shard (views 3..4 of 5) vs the single-process model vs a float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.dist import SequenceSubset, ShardPlan, slice_state
from nemo_cvpr2023_amd.neural_motion_model import NemoV3, ShardInfo, make_init_state
from oracle.model import OracleNemo
from test_dist import _args, _draws, V, T, NV
from test_gpu_model import _float64_twin

args = _args(3)
seqs = syn.SyntheticSequences(V, T, seed=1234)
torch.manual_seed(0)
state = make_init_state(args, 3, V, seqs.IMG_D0)
state['learned_motion.rot_out.weight'] = state['learned_motion.rot_out.weight'] * 2e3
kw = dict(smpl_assets=syn.make_smpl_assets(NV, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
m = NemoV3(args, seqs, 'cuda:0', **kw)
m.load_state_dict(state, strict=False)
plan = ShardPlan(V, T, 1, 2)
ms = NemoV3(args, SequenceSubset(seqs, plan.lo, plan.hi), 'cuda:0', **kw)
ms.load_state_dict(slice_state(state, plan.lo, plan.hi), strict=False)
o = OracleNemo(3, args, seqs, kw['smpl_assets'], kw['vposer_state'], kw['gmm'], state=state)
vi, fi = _draws(1)[0]
o64 = _float64_twin(o)
torch.set_default_dtype(torch.float64)
o64.step(vi, fi)
torch.set_default_dtype(torch.float32)
o.step(vi, fi)
m.step(vi, fi)
lv, lf, d = plan.route(vi, fi)
print('global batch views', vi.tolist(), 'local', lv.tolist(), d)
ms.step(lv, lf, _shard=ShardInfo(kr=d['kr'], mr=d['mr'], vr=d['vr'], n_global=d['n_global']))
nm, ns = dict(m.named_parameters()), dict(ms.named_parameters())
for i in range(plan.lo, plan.hi):
    for t in ('shifts', 'scales'):
        k = f'phase_networks.{i}.{t}'
        g64 = o64.P[k].grad.numpy()
        go = o.P[k].grad.numpy().astype(np.float64)
        gm = nm[k].grad.cpu().numpy().astype(np.float64)
        gs = ns[f'phase_networks.{i - plan.lo}.{t}'].grad.cpu().numpy().astype(np.float64)
        sc = np.abs(g64).max()
        print(k, 'scale %.3e' % sc, 'oracle32 %.2e' % (np.abs(go - g64).max() / sc), 'single %.2e' % (np.abs(gm - g64).max() / sc),
              'shard %.2e' % (np.abs(gs - g64).max() / sc), 'shard-vs-single %.2e' % (np.abs(gs - gm).max() / sc))
for k in ('learned_cameras', 'learned_instance_code'):
    g64 = o64.P[k].grad.numpy()[plan.lo:plan.hi]
    gm = nm[k].grad.cpu().numpy()[plan.lo:plan.hi]
    gs = ns[k].grad.cpu().numpy()
    sc = np.abs(g64).max()
    print(k, 'single %.2e shard %.2e' % (np.abs(gm - g64).max() / sc, np.abs(gs - g64).max() / sc))
