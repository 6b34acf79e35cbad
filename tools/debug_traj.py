"""Debug helper: per-step relative errors of the HIP trajectory vs the reference golden."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import rel_err
from test_gpu_model import build_hip_case
from test_oracle_golden import CASES, build_case

for name in sys.argv[1:]:
    m, g = build_hip_case(name)
    o, _, _ = build_case(name)
    V, Tn, B = int(g['meta__V']), int(g['meta__T']), int(g['meta__B'])
    torch.manual_seed(2)
    draw = lambda: (torch.randint(0, V, size=(B,)), torch.randint(0, Tn, size=(B,)))
    vi, fi = draw(); m.step(vi, fi, update=False, full_batch=True); o.step(vi, fi, update=False, full_batch=True)
    draw()
    if 'warmup_losses' in g:
        st = torch.get_rng_state(); m.warmup(len(g['warmup_losses'])); torch.set_rng_state(st); o.warmup(len(g['warmup_losses']))
    if 'cam_losses' in g:
        n = max(len(g['cam_losses']), 2)
        st = torch.get_rng_state(); m.opt_cam(n); torch.set_rng_state(st); o.opt_cam(n)
    for s in range(g['batches_view'].shape[0]):
        vi, fi = torch.as_tensor(g['batches_view'][s]), torch.as_tensor(g['batches_frame'][s])
        ld, info = m.step(vi, fi)
        ldo, infoo = o.step(vi, fi)
        tag = f'step{s}'
        named = dict(m.named_parameters())
        gerr = max(rel_err(named[k].grad, o.P[k].grad) for k in o.P if k != 'learned_betas' and o.P[k].grad is not None and float(o.P[k].grad.abs().max()) > 0)
        print(name, s, 'total vs ref %.2e' % rel_err(ld['total_loss'], g[tag + '__total_loss']),
              'loss_all vs ref %.2e' % rel_err(info['loss_all'], g[tag + '__loss_all']),
              '| oracle-vs-ref loss_all %.2e' % rel_err(infoo['loss_all'], g[tag + '__loss_all']),
              '| max grad err vs oracle %.2e' % gerr)
