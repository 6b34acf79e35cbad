#!/usr/bin/env python3
"""FK kernels inside a replayed graph: microseconds per launch (NEMO_HIP_LIB selects probe builds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import hipops as H
from nemo_cvpr2023_amd import _lib, synthetic as syn
from nemo_cvpr2023_amd.engine import SmplContext
L = _lib.load()
assets = syn.make_smpl_assets(128, seed=1)
jm = [int(x) for x in assets['joint_map']]
ctx = SmplContext(assets, [jm[i] for i in [38] + list(range(1, 25))], 'cuda:0')
for rows in (300, 600, 2400, 4800):
    R = torch.eye(3, device='cuda').reshape(1, 1, 9).repeat(rows, 24, 1).contiguous()
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF, dA, dR = Z(rows, 24, 12), Z(rows, 24, 3), Z(rows, 208), Z(rows, 24, 12), Z(rows, 24, 9)
    def fwd():
        L.nemo_fk_fwd(ctx.handle, rows, R.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st())
    def bwd():
        L.nemo_fk_bwd(ctx.handle, rows, R.data_ptr(), A.data_ptr(), dA.data_ptr(), None, PF.data_ptr(), 208, dR.data_ptr(), H.st())
    out = []
    for fn in (fwd, bwd):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): fn()
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        out.append(best / 20 * 1e6)
    print(f'rows {rows}: fk_fwd {out[0]:.1f} us  fk_bwd {out[1]:.1f} us   (per launch, 20 launches in one graph; host sync included)')
