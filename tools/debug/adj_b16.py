"""The bf16 blend-shape adjoint (nemo_gemm_bf16mem, N = 207, K = 20 670) per launch, and which plan ran (NEMO_GEMM_DEBUG=1).
usage: python tools/debug/adj_b16.py [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib
L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(16 << 20, device='cuda')
K, Kp = 20670, 20688
big = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)
for M in [int(a) for a in sys.argv[1:]] or [8192]:
    A = torch.randn(M, Kp, device='cuda').to(torch.bfloat16); B = torch.randn(207, Kp, device='cuda').to(torch.bfloat16)
    C = torch.zeros(M, 208, device='cuda')
    fn = lambda: L.nemo_gemm_bf16mem(M, 207, K, A.data_ptr(), Kp, B.data_ptr(), Kp, C.data_ptr(), 208, None, 0, None, 0, 0, 1.0, 0,
                                     None, 0, None, 0, None, 0, ws.data_ptr(), ws.numel() * 4, st)
    for _ in range(3): assert fn() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ref = A[:, :K].double() @ B[:, :K].double().T
    print(f'M={M:6d}: {sorted(ts)[len(ts) // 2]:7.1f} us per launch (cold operands) = {2e-6 * M * 207 * K / sorted(ts)[len(ts) // 2]:6.1f} TFLOP/s, err {float((C[:, :207].double() - ref).norm() / ref.norm()):.1e}')
