"""A/B of the blend-shape adjoint: nemo_gemm_f32 TT (blend shapes through LDS-DMA) against nemo_blend_adjoint (blend shapes
from the body model's MFMA-ordered image straight into registers).  usage: python tools/debug/adjoint_ab.py [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib, synthetic as syn
from nemo_cvpr2023_amd.engine import SmplContext

L = _lib.load()
assets = syn.make_smpl_assets(6890, seed=1, skin_nnz=4)
jm = [int(x) for x in assets['joint_map']]
ctx = SmplContext(assets, [jm[i] for i in [38] + list(range(1, 25))], 'cuda:0')
NV3 = 3 * 6890
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(16 << 20, device='cuda')
big = torch.empty(256 << 20, device='cuda', dtype=torch.uint8)
for M in [int(a) for a in sys.argv[1:]] or [2400]:
    lda = (M + 15) // 16 * 16
    A = torch.randn(3 * ctx.NVp, lda, device='cuda')
    ref = A[:NV3, :M].T.double() @ assets['posedirs'].double().cuda().T
    res = {}
    for name in ('gemm', 'image'):
        C = torch.zeros(M, 208, device='cuda')
        def run():
            if name == 'gemm':
                rc = L.nemo_gemm_f32(1, 1, M, 207, NV3, A.data_ptr(), lda, ctx.posedirs, ctx.ldP, C.data_ptr(), 208, None, 0, None,
                                     0, 0, 1.0, 0, 0, ws.data_ptr(), ws.numel() * 4, st)
            else:
                rc = L.nemo_blend_adjoint(M, NV3, A.data_ptr(), lda, ctx.posedirs_adj, ctx.posedirs_adj_bytes, C.data_ptr(), 208, 1.0,
                                          0, ws.data_ptr(), ws.numel() * 4, st)
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            big.zero_()                       # dVP^T comes from HBM in the step (the mesh kernel has just written 198 MB)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        err = float((C[:, :207].double() - ref).norm() / ref.norm())
        res[name] = (sorted(ts)[len(ts) // 2], err)
    print(f'M={M:6d}  TT through LDS {res["gemm"][0]:7.1f} us (err {res["gemm"][1]:.1e})   image {res["image"][0]:7.1f} us (err {res["image"][1]:.1e})')
