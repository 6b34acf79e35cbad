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
    ('big4096_nt', 0, 1, 4096, 4096, 4096, 1),
    ('big4096_nn', 0, 0, 4096, 4096, 4096, 1),
    ('big4096_tn', 1, 0, 4096, 4096, 4096, 1),
]
SWEEP = '--sweep' in sys.argv
if SWEEP:
    sys.argv.remove('--sweep')
if len(sys.argv) > 1:           # bench_gemm.py [--sweep] <name-substring>[,<name-substring>...] [reps] [M-override]
    keys = sys.argv[1].split(',')
    SHAPES = [s_ for s_ in SHAPES if any(k in s_[0] for k in keys)]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if len(sys.argv) > 3:
    SHAPES = [(n_, ta_, tb_, int(sys.argv[3]) if M_ >= 2400 else M_, N_, int(sys.argv[3]) if K_ == 2401 else K_, sp_)
              for n_, ta_, tb_, M_, N_, K_, sp_ in SHAPES]
g = torch.Generator().manual_seed(0)
for name, ta, tb, M, N, K, split in SHAPES:
    pad = lambda n: (n + 3) // 4 * 4
    A = H.dev(torch.randn((K, pad(M)) if ta else (M, pad(K)), generator=g))[:, :(M if ta else K)]
    B = H.dev(torch.randn((N, pad(K)) if tb else (K, pad(N)), generator=g))[:, :(K if tb else N)]
    C = torch.zeros(M, pad(N), device='cuda')[:, :N]
    def run(split_k):
        for _ in range(3):
            H.gemm(A, B, ta, tb, split_k=split_k, C=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            H.gemm(A, B, ta, tb, split_k=split_k, C=C)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / REPS
    if SWEEP:
        os.environ.pop('NEMO_GEMM_TILE', None)
    us = run(0)
    line = f'{name:20s} M={M:5d} N={N:5d} K={K:5d} auto {us:7.1f} us {2.0*M*N*K/us/1e6:6.1f} TF'
    if SWEEP:       # forced (tile, split) grid: calibration data for the host cost model in gemm.hip
        for tile in (64, 128):
            os.environ['NEMO_GEMM_TILE'] = str(tile)
            line += f'\n    tile{tile:3d}:'
            for sp in (1, 2, 3, 4, 5, 6, 8, 12, 16):
                if sp > 1 and (K + 31) // 32 // sp < 1:
                    break
                try:
                    line += f' s{sp}={run(sp):.1f}'
                except Exception:
                    line += f' s{sp}=n/a'
        os.environ.pop('NEMO_GEMM_TILE', None)
    print(line, flush=True)
