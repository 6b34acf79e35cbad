"""First MotionNet layer (nn.Linear(105, 1000)): forward and dX products with the weight at its natural row stride (105: rows
not 16-byte aligned -> first-generation kernel) against a copy padded to 108.  usage: python tools/debug/first_layer_pad.py [M]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib
L = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2401
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(16 << 20, device='cuda')
X = torch.randn(M, 108, device='cuda'); X[:, 105:] = 0
W = torch.randn(1000, 105, device='cuda')
Wp = torch.zeros(1000, 108, device='cuda'); Wp[:, :105] = W
b = torch.randn(1000, device='cuda')
H = torch.zeros(M, 1000, device='cuda'); dH = torch.randn(M, 1000, device='cuda'); dX = torch.zeros(M, 108, device='cuda')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = lambda *a: (lambda: L.nemo_gemm_f32(*a, ws.data_ptr(), ws.numel() * 4, st))
P = lambda x: x.data_ptr()
for name, w, ldw, K in (('ld 105', W, 105, 105), ('ld 108, K 105', Wp, 108, 105), ('ld 108, K 108', Wp, 108, 108)):
    f = t(g(0, 1, M, 1000, K, P(X), 108, P(w), ldw, P(H), 1000, P(b), 1, None, 0, 0, 1.0, 0, 0))
    ref = torch.relu(X[:, :105].double() @ W.double().T + b.double())
    e1 = float((H.double() - ref).norm() / ref.norm())
    d = t(g(0, 0, M, K, 1000, P(dH), 1000, P(w), ldw, P(dX), 108, None, 0, None, 0, 0, 1.0, 0, 0))
    e2 = float((dX[:, :105].double() - dH.double() @ W.double()).norm() / (dH.double() @ W.double()).norm())
    gw = torch.zeros(1000, ldw, device='cuda')
    p = t(g(1, 0, 1000, K, M, P(dH), 1000, P(X), 108, P(gw), ldw, None, 0, None, 0, 0, 1.0, 0, 0))
    print(f'{name:14s}: forward {f:6.1f} us (err {e1:.1e})   dX {d:6.1f} us (err {e2:.1e})   dW {p:6.1f} us')
