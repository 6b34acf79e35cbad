import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, hipops as H
g = torch.Generator().manual_seed(0)
for M, N in [(2401, 1000), (2400, 512), (2401, 148), (301, 1000)]:
  for K in [32, 64, 128, 256, 512]:
    A = H.dev(torch.randn(M, K, generator=g)); B = H.dev(torch.randn(N, K, generator=g)); C = torch.zeros(M, (N + 3) // 4 * 4, device='cuda')[:, :N]
    for _ in range(3): H.gemm(A, B, 0, 1, split_k=0, C=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): H.gemm(A, B, 0, 1, split_k=0, C=C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f'M={M} N={N} K={K}: {us:.1f} us')
# empty kernel launch floor
x = torch.zeros(64, device='cuda')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): H.gemm(x[:1].view(1,1), x[:1].view(1,1), 0, 1, split_k=1, C=x[1:2].view(1,1))
e1.record(); torch.cuda.synchronize()
print('1x1x1 gemm launch:', e0.elapsed_time(e1) * 1e3 / 50, 'us')
