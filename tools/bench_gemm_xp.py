"""us per launch of nemo_gemm_xp against nemo_gemm_f32 for the MotionNet chain's products (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, '.')
from tests import hipops as H

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2401


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, N, K, kw in [('fwd hidden (copies)', rows, 1000, 1000, dict(want_cx=True, want_cxt=True, act=1)),
                          ('fwd hidden (C)', rows, 1000, 1000, {}),
                          ('dX hidden (mask, copies, colsum)', rows, 1000, 1000, dict(want_cx=True, want_cxt=True, colsum=True, mask=True)),
                          ('fwd first', rows, 1000, 105, dict(want_cx=True, want_cxt=True, act=1)),
                          ('fwd head', rows, 147, 1000, {}),
                          ('dX head', rows, 1000, 147, dict(want_cx=True, want_cxt=True, colsum=True, mask=True)),
                          ('dW hidden', 1000, 1000, rows, dict(out_mode=1)),
                          ('dW head', 147, 1000, rows, dict(out_mode=1)),
                          ('dW first', 1000, 105, rows, dict(out_mode=1)),
                          ('dX first', rows, 105, 1000, {}),
                          ('blend-shape adjoint', rows - 1, 207, 20670, dict(out_mode=1))]:
    A = torch.randn(M, K, device=H.DEV)
    B = torch.randn(N, K, device=H.DEV)
    mA = H.absmax_meta(A)[0] if fmt == 2 else None
    mB = H.absmax_meta(B)[0] if fmt == 2 else None
    mo = torch.zeros(64, device=H.DEV)
    Ax, _ = H.cast_xp(fmt, A, meta=mA)
    Bx, _ = H.cast_xp(fmt, B, meta=mB)
    kw = dict(kw)
    maskx = None
    if kw.pop('mask', False):
        maskx, _ = H.cast_xp(fmt, torch.randn(M, N, device=H.DEV).clamp_min(0))
    C = torch.zeros(M, N, device=H.DEV) if not kw.get('want_cx') else None
    L = H._lib.load()
    Cx = torch.zeros(M, H.xp_ld(fmt, N), dtype=torch.int16, device=H.DEV) if kw.get('want_cx') else None
    CxT = torch.zeros(N, H.xp_ld(fmt, M), dtype=torch.int16, device=H.DEV) if kw.get('want_cxt') else None
    cs = torch.zeros(int(L.nemo_gemm_colsum_rows(M)), N, device=H.DEV) if kw.get('colsum') else None
    ws = H.gemm_ws()

    def run():
        H.check(L.nemo_gemm_xp(fmt, M, N, K, H.dptr(Ax), Ax.stride(0), H.dptr(Bx), Bx.stride(0), H.dptr(C), N, None, kw.get('act', 0),
                               H.dptr(maskx), maskx.stride(0) if maskx is not None else 0, 1 if maskx is not None else 0, 1.0,
                               kw.get('out_mode', 0), H.dptr(Cx), Cx.stride(0) if Cx is not None else 0, H.dptr(CxT),
                               CxT.stride(0) if CxT is not None else 0, 1.0, H.dptr(cs), N if cs is not None else 0, H.dptr(mA), H.dptr(mB), None,
                               H.dptr(mo) if Cx is not None else None, None, H.dptr(ws),
                               ws.numel() * 4, H.st()), 'xp')
    t_xp = timeit(run)
    Cf = torch.zeros(M, N, device=H.DEV)
    t_32 = timeit(lambda: H.gemm(A, B, 0, 1, C=Cf))
    gf = 2.0 * M * N * K / 1e9
    print(f'{name:36s} M={M:6d} N={N:5d} K={K:6d}  xp{fmt} {t_xp:7.1f} us ({gf / t_xp * 1e3:6.0f} TF alg)   f32 {t_32:7.1f} us ({gf / t_32 * 1e3:6.0f} TF)', flush=True)
