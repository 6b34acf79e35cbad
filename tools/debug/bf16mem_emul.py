import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
V, T = 4, 300
args = syn.published_args(batch_size=512, out_dir='')
args.gemm_dtype = 'bf16'
seqs = syn.SyntheticSequences(V, T, seed=1234)
kw = dict(smpl_assets=syn.make_smpl_assets(6890, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
torch.manual_seed(0)
m = NemoV2(args, seqs, 'cuda:0', **kw)
with torch.no_grad():
    m.learned_motion.rot_out.weight.mul_(2e3)
for o in m.optimizers:
    o.param_groups[0]['lr'] = 0.0
m.step(None, None, full_batch=True)
e = m.engine
N = V * T
w = e._ws(N)
r = N + 1
bf = lambda x: x.to(torch.bfloat16).double()
lm = m.learned_motion
X = w['X'][:r, :e.din]
L0, L2, L4 = getattr(lm.net.net, '0'), getattr(lm.net.net, '2'), getattr(lm.net.net, '4')
W0, W2, W4 = L0.weight.detach(), L2.weight.detach(), L4.weight.detach()
Wh = torch.cat([lm.rot_out.weight.detach(), lm.linear_out.weight.detach()], 0)
b0, b2, b4 = L0.bias.detach().double(), L2.bias.detach().double(), L4.bias.detach().double()
bh = torch.cat([lm.rot_out.bias.detach(), lm.linear_out.bias.detach()]).double()
rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
H1 = torch.relu(bf(X) @ bf(W0).T + b0); print('H1', rel(w['H1'][:r], H1))
H2 = torch.relu(bf(w['H1'][:r]) @ bf(W2).T + b2); print('H2', rel(w['H2'][:r], H2))
H3 = torch.relu(bf(w['H2'][:r]) @ bf(W4).T + b4); print('H3', rel(w['H3'][:r], H3))
HEAD = bf(w['H3'][:r]) @ bf(Wh).T + bh; print('HEAD', rel(w['HEAD'][:r, :147], HEAD))
dHEAD = w['dHEAD'][:r, :147]
dH = (bf(dHEAD) @ bf(Wh)) * (w['H3'][:r] > 0); print('dH', rel(w['dH'][:r], dH))
dHb = (bf(w['dH'][:r]) @ bf(W4)) * (w['H2'][:r] > 0); print('dH_b', rel(w['dH_b'][:r], dHb))
dHc = (bf(w['dH_b'][:r]) @ bf(W2)) * (w['H1'][:r] > 0); print('dH_c', rel(w['dH_c'][:r], dHc))
dX = bf(w['dH_c'][:r]) @ bf(W0); print('dX', rel(w['dX'][:r, :e.din], dX))
g = dict(m.named_parameters())
print('dW_head', rel(torch.cat([g['learned_motion.rot_out.weight'].grad, g['learned_motion.linear_out.weight'].grad], 0), bf(dHEAD).T @ bf(w['H3'][:r])))
print('dW4', rel(g['learned_motion.net.net.4.weight'].grad, bf(w['dH'][:r]).T @ bf(w['H2'][:r])))
print('dW2', rel(g['learned_motion.net.net.2.weight'].grad, bf(w['dH_b'][:r]).T @ bf(w['H1'][:r])))
print('dW0', rel(g['learned_motion.net.net.0.weight'].grad, bf(w['dH_c'][:r]).T @ bf(X)))
print('db2', rel(g['learned_motion.net.net.2.bias'].grad, w['dH_b'][:r].double().sum(0)))
