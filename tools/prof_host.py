import cProfile, pstats, sys, torch, time
sys.path.insert(0, '/root/repo')
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
args = syn.published_args(batch_size=512, out_dir='')
seqs = syn.SyntheticSequences(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 300, seed=1234)
m = NemoV2(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(6890, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
for _ in range(5): m.step(None, None, update=True, full_batch=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): m.step(None, None, update=True, full_batch=True)
print('ms/step', (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): m.step(None, None, update=True, full_batch=True)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
