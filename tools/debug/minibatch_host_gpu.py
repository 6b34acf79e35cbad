"""Random-minibatch steps (512 of 8 x 300, published configuration): wall clock per step against the GPU's own time per step
(HIP events around every step's launches) and the host time spent inside step() before / after the loss wait."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
args = syn.published_args(batch_size=512, out_dir='')
seqs = syn.SyntheticSequences(8, 300, seed=1234)
m = NemoV2(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(6890, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
gen = torch.Generator().manual_seed(1)
draws = [(torch.randint(0, 8, (512,), generator=gen), torch.randint(0, 300, (512,), generator=gen)) for _ in range(420)]
for vi, fi in draws[:20]:
    m.step(vi, fi)
e = m.engine
waits = []
orig = e.wait_scalars
def timed_wait(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); waits.append(time.perf_counter() - t0); return r
e.wait_scalars = timed_wait
evs = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for vi, fi in draws[20:]:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); m.step(vi, fi); b.record(); evs.append((a, b))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 400
gpu = sum(a.elapsed_time(b) for a, b in evs) / 400
print('wall %.4f ms/step; GPU start-to-end of a step %.4f ms; host waiting for the losses %.4f ms/step -> host busy %.4f ms/step'
      % (wall * 1e3, gpu, 1e3 * sum(waits) / 400, 1e3 * (wall - sum(waits) / 400)))
