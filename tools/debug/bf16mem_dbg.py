import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import hipops as H
from nemo_cvpr2023_amd import _lib
from nemo_cvpr2023_amd._lib import check, dptr
L = _lib.load()
ws = H.gemm_ws()
for M, N, K in ((64, 64, 64), (64, 64, 128), (64, 64, 16), (64, 64, 1000), (128, 128, 256)):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).cuda(); B = torch.randn(N, K, generator=g).cuda()
    Kp = (K + 7) // 8 * 8
    Ab = torch.zeros(M, Kp, dtype=torch.int16, device='cuda'); Bb = torch.zeros(N, Kp, dtype=torch.int16, device='cuda')
    check(L.nemo_cast_bf16(M, K, dptr(A), K, dptr(Ab), Kp, 0, H.st()), 'c'); check(L.nemo_cast_bf16(N, K, dptr(B), K, dptr(Bb), Kp, 0, H.st()), 'c')
    C = torch.zeros(M, N, device='cuda')
    check(L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), Kp, dptr(Bb), Kp, dptr(C), N, None, 0, None, 0, 0, 1.0, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st()), 'g')
    ref = A.bfloat16().double() @ B.bfloat16().double().T
    err = (C.double() - ref).abs()
    print(M, N, K, 'max err', float(err.max()), 'scale', float(ref.abs().max()), 'bad elems', int((err > 1e-3).sum()))
    if K <= 128:
        # which k contribute wrongly? probe with one-hot columns
        for k0 in range(0, K, 8):
            A1 = torch.zeros(M, K).cuda(); A1[:, k0:k0 + 8] = A[:, k0:k0 + 8]
            check(L.nemo_cast_bf16(M, K, dptr(A1), K, dptr(Ab), Kp, 0, H.st()), 'c')
            check(L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), Kp, dptr(Bb), Kp, dptr(C), N, None, 0, None, 0, 0, 1.0, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st()), 'g')
            r1 = A1.bfloat16().double() @ B.bfloat16().double().T
            e = float((C.double() - r1).abs().max())
            if e > 1e-3:
                print('   k block', k0, 'err', e)
print('---- variants at 301x1000x1000')
M, N, K = 301, 1000, 1000
g = torch.Generator().manual_seed(2)
A = torch.randn(M, K + 3, generator=g).cuda()[:, :K]; Bt = torch.randn(K, N + 1, generator=g).cuda()[:, :N]
bias = torch.randn(N, generator=g).cuda()
Ab = torch.zeros(M, K, dtype=torch.int16, device='cuda'); Bb = torch.zeros(N, K, dtype=torch.int16, device='cuda')
check(L.nemo_cast_bf16(M, K, dptr(A), A.stride(0), dptr(Ab), K, 0, H.st()), 'c'); check(L.nemo_cast_bf16(K, N, dptr(Bt), Bt.stride(0), dptr(Bb), K, 1, H.st()), 'c')
print('cast ok', torch.equal(Ab.view(torch.bfloat16), A.bfloat16()), torch.equal(Bb.view(torch.bfloat16), Bt.T.bfloat16()))
pre = A.bfloat16().double() @ Bt.bfloat16().double()
Mp = 304
for use_bias, act, cb in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1)):
    C = torch.zeros(M, N, device='cuda'); Cb = torch.zeros(M, N, dtype=torch.int16, device='cuda'); CbT = torch.zeros(N, Mp, dtype=torch.int16, device='cuda')
    check(L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), K, dptr(Bb), K, dptr(C), N, dptr(bias) if use_bias else None, act, None, 0, 0, 1.0, 0,
                              dptr(Cb) if cb else None, N, dptr(CbT) if cb else None, Mp, dptr(ws), ws.numel() * 4, H.st()), 'g')
    ref = pre + (bias.double() if use_bias else 0)
    if act: ref = torch.relu(ref)
    err = (C.double() - ref).abs()
    print('bias', use_bias, 'act', act, 'cb', cb, 'max err', float(err.max()), 'rel', float(err.max() / ref.abs().max()), 'where', divmod(int(err.argmax()), N))
