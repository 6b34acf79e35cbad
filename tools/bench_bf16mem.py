"""nemo_gemm_bf16mem against nemo_gemm_bf16 / nemo_gemm_f32 at the C3 shapes (N = 12 000): microseconds per launch (HIP events, 20 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import hipops as H
from nemo_cvpr2023_amd import _lib
from nemo_cvpr2023_amd._lib import check, dptr
L = _lib.load()
ws = H.gemm_ws()


def timeit(fn, n=20):
    for _ in range(3):
        assert fn() == 0, 'launch failed'
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for M, N, K in ((12000, 1000, 1000), (2401, 1000, 1000), (12000, 512, 512), (12000, 147, 1000), (1000, 1000, 12000), (6000, 208, 20672)):
    A, B = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
    C = torch.zeros(M, N, device='cuda')
    Ab, Bb = torch.zeros(M, K, dtype=torch.int16, device='cuda'), torch.zeros(N, K, dtype=torch.int16, device='cuda')
    Cb, CbT = torch.zeros(M, N, dtype=torch.int16, device='cuda'), torch.zeros(N, (M + 7) // 8 * 8, dtype=torch.int16, device='cuda')
    check(L.nemo_cast_bf16(M, K, dptr(A), K, dptr(Ab), K, 0, H.st()), 'c')
    check(L.nemo_cast_bf16(N, K, dptr(B), K, dptr(Bb), K, 0, H.st()), 'c')
    f32 = lambda: L.nemo_gemm_f32(0, 1, M, N, K, dptr(A), K, dptr(B), K, dptr(C), N, None, 1, None, 0, 0, 1.0, 0, 0, dptr(ws), ws.numel() * 4, H.st())
    b16 = lambda: L.nemo_gemm_bf16(0, 1, M, N, K, dptr(A), K, dptr(B), K, dptr(C), N, None, 1, None, 0, 0, 1.0, 0, 0, dptr(ws), ws.numel() * 4, H.st())
    mem = lambda: L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), K, dptr(Bb), K, dptr(C), N, None, 1, None, 0, 0, 1.0, 0, None, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st())
    cs = torch.zeros(2 * ((M + 63) // 64), N, device='cuda:0')
    mem3 = lambda: L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), K, dptr(Bb), K, None, 0, None, 1, None, 0, 0, 1.0, 0, dptr(Cb), N, dptr(CbT), CbT.stride(0), dptr(cs), N, dptr(ws), ws.numel() * 4, H.st())
    mem2 = lambda: L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), K, dptr(Bb), K, dptr(C), N, None, 1, None, 0, 0, 1.0, 0, dptr(Cb), N, dptr(CbT), CbT.stride(0), None, 0, dptr(ws), ws.numel() * 4, H.st())
    cast = lambda: L.nemo_cast_bf16(M, K, dptr(A), K, dptr(Ab), K, 0, H.st())
    castT = lambda: L.nemo_cast_bf16(M, N, dptr(C), N, dptr(CbT), CbT.stride(0), 1, H.st())
    gf = 2e-9 * M * N * K
    t = [timeit(f) for f in (f32, b16, mem, mem2, cast, castT, mem3)]
    print(f'{M}x{N}x{K} ({gf:.1f} GFLOP): f32 {t[0]:.1f} us ({gf / t[0] * 1e3:.0f} TF)  bf16 on the fly {t[1]:.1f} ({gf / t[1] * 1e3:.0f} TF)  '
          f'bf16 in memory {t[2]:.1f} ({gf / t[2] * 1e3:.0f} TF)  + Cb/CbT {t[3]:.1f}  Cb/CbT + column sums, no fp32 C {t[6]:.1f}  | cast A {t[4]:.1f}  castT C {t[5]:.1f}')
