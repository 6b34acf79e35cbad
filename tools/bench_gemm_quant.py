import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, hipops as H
os.environ['NEMO_GEMM_TILE'] = '64'
g = torch.Generator().manual_seed(0)
for M, N, K in [(2048,1024,1024),(2048,1024,4096),(2304,1024,1024),(2401,1000,1000),(2560,1024,1024),(3072,1024,1024),(4096,1024,1024),(1024,1024,1024),(4096,4096,1024)]:
    A = H.dev(torch.randn(M, K, generator=g)); B = H.dev(torch.randn(N, K, generator=g)); C = torch.zeros(M, N, device='cuda')
    for _ in range(3): H.gemm(A, B, 0, 1, split_k=1, C=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): H.gemm(A, B, 0, 1, split_k=1, C=C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    print(f'M={M} N={N} K={K} tiles={tiles} ({tiles/256:.2f}/CU) {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TF')
