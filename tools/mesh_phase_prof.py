import os, sys, ctypes
os.environ['NEMO_HIP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'nemo_cvpr2023_amd', 'libnemo_hip_abl.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_cvpr2023_amd import synthetic as syn, _lib
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = syn.published_args(batch_size=512, out_dir='')
if len(sys.argv) > 2:
    args.gemm_dtype = sys.argv[2]
seqs = syn.SyntheticSequences(V, 300, seed=1234)
nnz = int(os.environ.get('SKIN_NNZ', '24'))
m = NemoV2(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(int(os.environ.get('NUM_VERTS', '6890')), seed=1, skin_nnz=nnz), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
if os.environ.get('SKIN_DENSE'): m.engine.ctx.set_skin_sparse(False)
for _ in range(6): m.step(None, None, update=True, full_batch=True)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['NEMO_HIP_LIB'])
out = (ctypes.c_ulonglong * 8192)()
print('rc', L.nemo_debug_mesh_prof(out))
a = np.array(out[:], dtype=np.float64).reshape(1024, 8)
print('blk: tiles | per tile: blend (ideal alone 9984) / rec (2304) / orig+adj+store (5376) / sum (17664) || of the third phase: orig skinning MFMAs (2304) / sign+dvp VALU / adjoint (3072) / stores')
for b in list(range(0, 512, 32)) + [255, 256, 449, 450, 499]:
    bl, rec, org, nt, sk, va, ad, stt = a[b]
    if nt == 0: continue
    print(f'{b:4d}: {int(nt):3d} | {bl/nt:6.0f} {rec/nt:6.0f} {org/nt:6.0f} {(bl+rec+org)/nt:6.0f} || {sk/nt:6.0f} {va/nt:6.0f} {ad/nt:6.0f} {stt/nt:6.0f}')
