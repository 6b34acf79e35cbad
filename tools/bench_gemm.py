#!/usr/bin/env python3
"""Micro-benchmark of nemo_gemm_f32 on the shapes of the NeMo step (HIP events, 20 launches each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import hipops as H

SHAPES = [  # name, ta, tb, M, N, K, split
    ('mlp_hidden_fwd', 0, 1, 2401, 1000, 1000, 1),
    ('mlp_hidden_dx', 0, 0, 2401, 1000, 1000, 1),
    ('mlp_hidden_dw', 1, 0, 1000, 1000, 2401, 2),
    ('mlp_in_fwd(K=105)', 0, 1, 2401, 1000, 105, 1),
    ('rot_out_fwd', 0, 1, 2401, 144, 1000, 1),
    ('rot_out_dx', 0, 0, 2401, 1000, 144, 1),
    ('vposer_512', 0, 1, 2400, 512, 512, 1),
    ('vposer_in(K=63)', 0, 1, 2400, 512, 63, 1),
    ('vposer_mulv', 0, 1, 2400, 64, 512, 1),
    ('vposer_dec_out', 0, 1, 2400, 126, 512, 1),
    ('mq', 0, 0, 2400, 792, 207, 1),
    ('dpf_kp', 0, 1, 2400, 207, 792, 1),
    ('pose_blend_bwd_TT', 1, 1, 2400, 207, 20670, 8),
    ('pose_blend_fwd', 0, 0, 4800, 20670, 207, 1),
]
g = torch.Generator().manual_seed(0)
for name, ta, tb, M, N, K, split in SHAPES:
    pad = lambda n: (n + 3) // 4 * 4
    A = H.dev(torch.randn((K, pad(M)) if ta else (M, pad(K)), generator=g))[:, :(M if ta else K)]
    B = H.dev(torch.randn((N, pad(K)) if tb else (K, pad(N)), generator=g))[:, :(K if tb else N)]
    C = torch.zeros(M, pad(N), device='cuda')[:, :N]
    for _ in range(3):
        H.gemm(A, B, ta, tb, out_mode=2 if split > 1 else 0, split_k=split, C=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.gemm(A, B, ta, tb, out_mode=2 if split > 1 else 0, split_k=split, C=C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'{name:24s} M={M:5d} N={N:5d} K={K:5d} split={split}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s')
