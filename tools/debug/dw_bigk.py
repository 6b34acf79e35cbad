"""Parameter-gradient product dW = dY^T X of a hidden layer at C4's batch (K = 262 145 rows): us per launch and the plan the
library picks (NEMO_GEMM_DEBUG=1); NEMO_GEMM_TILE=128 forces the 128 x 128 tile.  usage: python tools/debug/dw_bigk.py [K]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib
L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(16 << 20, device='cuda')
P = lambda x: x.data_ptr()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 262145
dY = torch.randn(K, 1000, device='cuda'); X = torch.randn(K, 1000, device='cuda'); G = torch.zeros(1000, 1000, device='cuda')
fn = lambda: L.nemo_gemm_f32(1, 0, 1000, 1000, K, P(dY), 1000, P(X), 1000, P(G), 1000, None, 0, None, 0, 0, 1.0, 1, 0, ws.data_ptr(), ws.numel() * 4, st)
for _ in range(2): assert fn() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print(f'K={K}: {us:.0f} us per launch = {2e-6 * 1000 * 1000 * K / us:.1f} TFLOP/s  (tile override: {os.environ.get("NEMO_GEMM_TILE", "-")})')
