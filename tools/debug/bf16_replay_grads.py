"""bf16-in-memory chain: gradients of captured / replayed steps against eager steps (same state), small sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV3

def build(V, graphs):
    over = dict(h_dim=16, monotonic_network_n_nodes=10, batch_size=12, out_dir='', phase_rbf_dim=8, weight_instance_loss=0.1, weight_3d_loss=0.5)
    args = syn.published_args(**over); args.gemm_dtype = 'bf16'
    seqs = syn.SyntheticSequences(V, 6, seed=1234)
    torch.manual_seed(0)
    m = NemoV3(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(64, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    m.use_graphs = graphs
    with torch.no_grad():
        m.learned_motion.rot_out.weight.mul_(2e3)
    return m

for V in (5, 3, 2):
    a, b = build(V, True), build(V, False)
    for it in range(5):
        sd = {k: v.detach().clone() for k, v in a.state_dict().items()}
        b.load_state_dict(sd, strict=False)
        a.step(None, None, full_batch=True); b.step(None, None, full_batch=True)
        torch.cuda.synchronize()
        worst = (0, '')
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            if p.grad is None: continue
            d = float((p.grad - q.grad).abs().max()); s = float(q.grad.abs().max())
            if s > 0 and d / s > worst[0]: worst = (d / s, k)
        print(f'V={V} step {it}: worst relative gradient difference graph vs eager {worst[0]:.3g} ({worst[1]})')
