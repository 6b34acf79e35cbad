"""C3 bf16: parameter gradients of one full-batch step against the fp32 HIP path, first layer on the bf16 chain or in fp32
(NEMO_B16_FIRST_LAYER=1 / 0): cosine and largest entry error per tensor (what tests/test_gpu_bf16.py gates)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import torch
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
DEV = 'cuda:0'
V, T = 40, 300
seqs = syn.SyntheticSequences(V, T, seed=1234)
assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=4), syn.make_vposer_state(), syn.make_gmm()
models = {}
for dt in ('f32', 'bf16'):
    args = syn.published_args(batch_size=512, out_dir='')
    args.gemm_dtype = dt
    torch.manual_seed(0)
    m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    for o_ in m.optimizers:
        o_.param_groups[0]['lr'] = 0.0
    m.step(None, None, update=True, full_batch=True)
    models[dt] = m
n32, n16 = dict(models['f32'].named_parameters()), dict(models['bf16'].named_parameters())
print('B16_FIRST_LAYER', models['bf16'].engine.B16_FIRST_LAYER)
for k in n32:
    if n32[k].grad is None or n16[k].grad is None:
        continue
    a, b = n16[k].grad.double().flatten(), n32[k].grad.double().flatten()
    if float(b.norm()) == 0:
        continue
    cos = float(a @ b / (a.norm() * b.norm()))
    err = float((a - b).abs().max() / b.abs().max())
    print(f'{k:50s} cos {cos:.5f}  max entry err {err:.4f}  norm ratio {float(a.norm() / b.norm()):.4f}')
