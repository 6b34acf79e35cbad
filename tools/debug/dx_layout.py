"""Activation-gradient product dX = (dY W) * mask of a hidden layer: W as stored (out x in: "NN", the B operand is not
k-contiguous) against a transposed copy ("NT", both operands k-contiguous).  usage: python tools/debug/dx_layout.py [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib
L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(16 << 20, device='cuda')
P = lambda x: x.data_ptr()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in [int(a) for a in sys.argv[1:]] or [2401]:
    dY = torch.randn(M, 1000, device='cuda'); W = torch.randn(1000, 1000, device='cuda'); WT = W.t().contiguous()
    H = torch.randn(M, 1000, device='cuda'); out = torch.zeros(M, 1000, device='cuda'); out2 = torch.zeros(M, 1000, device='cuda')
    nn = lambda: L.nemo_gemm_f32(0, 0, M, 1000, 1000, P(dY), 1000, P(W), 1000, P(out), 1000, None, 0, P(H), 1000, 1, 1.0, 0, 0, ws.data_ptr(), ws.numel() * 4, st)
    nt = lambda: L.nemo_gemm_f32(0, 1, M, 1000, 1000, P(dY), 1000, P(WT), 1000, P(out2), 1000, None, 0, P(H), 1000, 1, 1.0, 0, 0, ws.data_ptr(), ws.numel() * 4, st)
    a, b = t(nn), t(nt)
    tr = t(lambda: WT.copy_(W.t()))
    print(f'M={M:7d}  NN {a:8.1f} us   NT {b:8.1f} us   (transpose of W {tr:.1f} us)   max diff {float((out - out2).abs().max()):.2e}')
