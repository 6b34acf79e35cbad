import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_gpu_model import build_hip_case
from test_oracle_golden import build_case
torch.set_printoptions(precision=4, linewidth=200, sci_mode=True)
name='v3_small'
m, g = build_hip_case(name); o,_,_ = build_case(name)
V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
torch.manual_seed(3)
named = dict(m.named_parameters())
for s in range(7):
    o.load_state({k: v.cpu() for k, v in m.state_dict().items()})
    for oo, mo in zip(o.optimizers, m.optimizers):
        sd = mo.state_dict()
        if sd['state']: oo.load_state_dict(sd)
    vi, fi = torch.randint(0, V, (B,)), torch.randint(0, Tn, (B,))
    # float64 truth of the phase-net gradient on the same state
    if s == 6:
        import copy
        o64 = copy.deepcopy(o)
        torch.set_default_dtype(torch.float64)
        for k in o64.P: o64.P[k] = o64.P[k].detach().double().requires_grad_(True)
        for attr in ('points2d_gt_all','gt_bbox_size','hmr_theta','hmr_mask','rbf_centres'):
            setattr(o64, attr, getattr(o64, attr).double())
        for k in o64.smpl.a:
            if isinstance(o64.smpl.a[k], torch.Tensor) and o64.smpl.a[k].is_floating_point(): o64.smpl.a[k] = o64.smpl.a[k].double()
        o64.vp.sd = {k: v.double() for k, v in o64.vp.sd.items()}
        o64.prior.means = o64.prior.means.double(); o64.prior.precisions = o64.prior.precisions.double(); o64.prior.nll_weights = o64.prior.nll_weights.double()
        o64._build_optimizers()
        o64.step(vi, fi)
        torch.set_default_dtype(torch.float32)
    m.step(vi, fi); o.step(vi, fi)
print('vi',vi,'fi',fi)
for k in ('phase_networks.0.shifts','phase_networks.0.scales','phase_networks.1.shifts'):
    print(k); print(' hip', named[k].grad.cpu()); print(' orc', o.P[k].grad); print(' f64', o64.P[k].grad.float())
