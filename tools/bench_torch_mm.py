import torch
for (M,N,K) in [(2400,1000,1000),(300,1000,1000),(1000,1000,2400),(2400,207,20670),(4096,4096,4096),(2400,512,512)]:
    A=torch.randn(M,K,device='cuda'); B=torch.randn(N,K,device='cuda')
    for _ in range(3): C=A@B.T
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C=A@B.T
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    print(f'torch fp32 NT {M}x{N}x{K}: {us:.1f} us {2*M*N*K/us/1e6:.1f} TF')
