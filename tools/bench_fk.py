#!/usr/bin/env python3
"""Scaling probe for the FK kernels (run under rocprofv3 --kernel-trace)."""
import os, sys
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
for rows in (64, 640, 2400, 4800, 38400):
    R = torch.eye(3, device='cuda').reshape(1, 1, 9).repeat(rows, 24, 1).contiguous()
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF, dA, dR = Z(rows, 24, 12), Z(rows, 24, 3), Z(rows, 208), Z(rows, 24, 12), Z(rows, 24, 9)
    for _ in range(5):
        L.nemo_fk_fwd(ctx.handle, rows, R.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st())
        L.nemo_fk_bwd(ctx.handle, rows, R.data_ptr(), A.data_ptr(), dA.data_ptr(), None, PF.data_ptr(), 208,
                      dR.data_ptr(), H.st())
    torch.cuda.synchronize()
