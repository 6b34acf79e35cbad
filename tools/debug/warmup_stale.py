import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from test_gpu_model import _tiny_v2, _copy_model_state
from oracle.model import OracleNemo
a, (version, args, seqs, assets, vps, gmm) = _tiny_v2()
b, _ = _tiny_v2()
with torch.no_grad():
    a.learned_motion.rot_out.weight.mul_(2e3)
torch.manual_seed(1)
a.warmup(1)
_copy_model_state(b, a)
o = OracleNemo(version, args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in a.state_dict().items()})
args.batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for mdl in (a, b, o):
    torch.manual_seed(2)
    print('loss', mdl.warmup(1))
na, nb = dict(a.named_parameters()), dict(b.named_parameters())
for k, p in o.P.items():
    if k == 'learned_betas':
        continue
    go = p.grad
    ga, gb = na[k].grad.cpu(), nb[k].grad.cpu()
    sc = float(ga.abs().max()) + 1e-30
    print('%-40s scale %.2e  a-b %.2e  a-o %s  b-o %s' % (k, sc, float((ga - gb).abs().max()) / sc,
          'none' if go is None else '%.2e' % (float((ga - go).abs().max()) / sc), 'none' if go is None else '%.2e' % (float((gb - go).abs().max()) / sc)))
