#!/usr/bin/env python3
"""Per-block wall-clock phases of the fused mesh kernel (library built by tools/mesh_timeline_build.sh).
   python tools/mesh_timeline.py [instances]   -- microseconds relative to the first block's entry."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['NEMO_HIP_LIB'] = os.path.join(ROOT, 'nemo_cvpr2023_amd', 'libnemo_hip_tl.so')
os.environ['NEMO_GRAPHS'] = '0'
sys.path.insert(0, ROOT)
import numpy as np, torch
from nemo_cvpr2023_amd import synthetic as syn
from nemo_cvpr2023_amd.neural_motion_model import NemoV2
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = syn.published_args(batch_size=512, out_dir='')
seqs = syn.SyntheticSequences(V, 300, seed=1234)
m = NemoV2(args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(6890, seed=1), vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
for _ in range(6):
    m.step(None, None, update=True, full_batch=True)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['NEMO_HIP_LIB'])
out = (ctypes.c_ulonglong * 8192)()
assert L.nemo_debug_mesh_timeline(out) == 0
a = np.array(out[:], dtype=np.float64).reshape(1024, 8)[:, :5]
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
a = (a - t0) / 100.0                       # 100 MHz -> microseconds
nb = len(a)
print(f'{nb} blocks; kernel span (first entry -> last end) {a[:, 4].max():.1f} us')
names = ['entry', 'staged', 'tiles done', 'combine done', 'end']
for q in range(5):
    print(f'  {names[q]:13s} min {a[:, q].min():7.1f}  median {np.median(a[:, q]):7.1f}  max {a[:, q].max():7.1f}')
d = np.diff(a, axis=1)
for q, nme in enumerate(['staging', 'tile loop', 'cross-wave + group combine', 'L1 grid reduction']):
    print(f'  phase {nme:28s} median {np.median(d[:, q]):7.1f}  max {d[:, q].max():7.1f} us')
late = a[:, 0] > 5.0
print(f'  blocks entering later than 5 us after the first: {int(late.sum())} (second wave of blocks / dispatch tail)')
full = np.array(out[:], dtype=np.float64).reshape(1024, 8)
full = full[full[:, 0] > 0]
last = full[full[:, 7] > 0]                    # blocks that summed a group's partials
if len(last):
    f = (last - t0) / 100.0
    print(f'  last arrivers ({len(last)}): cross-wave pass {np.median(f[:, 5] - f[:, 2]):5.1f} | partial store + ticket {np.median(f[:, 6] - f[:, 5]):5.1f} | '
          f'sum of the partials {np.median(f[:, 7] - f[:, 6]):5.1f} (max {np.max(f[:, 7] - f[:, 6]):5.1f}) | dA write {np.median(f[:, 3] - f[:, 7]):5.1f} us')
    oth = full[full[:, 7] == 0]
    g = (oth - t0) / 100.0
    print(f'  other blocks: cross-wave pass {np.median(g[:, 5] - g[:, 2]):5.1f} | partial store + ticket {np.median(g[:, 6] - g[:, 5]):5.1f} us')
if os.environ.get('TL_DUMP'):
    raw = np.array(out[:], dtype=np.float64).reshape(1024, 8)[:, :5]
    for b in range(len(raw)):
        if raw[b, 0] == 0: continue
        if b % int(os.environ['TL_DUMP']) == 0 or b >= nb - 8:
            r = (raw[b] - t0) / 100.0
            print(f'block {b:4d} (CU-pair slot {b % 256:3d}): entry {r[0]:6.1f} staged {r[1]:6.1f} tiles {r[2]:6.1f} combine {r[3]:6.1f} end {r[4]:6.1f}')
